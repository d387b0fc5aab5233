"""CPU tests (-m "not gpu") of bench.py's N > 1 launcher: `python bench.py --gpus 2` re-launches itself under
torch.distributed.run (two ranks, gloo on this GPU-less box through the hidden --plumbing-test switch) and runs the
same sharding / super-block broadcast / barrier / max-over-ranks timing / JSON code the GPU run uses.  The compute is a
no-op stand-in here -- it proves the launch plumbing, not two HIP engines (those need GPUs; the engine itself is covered
by the -m gpu tests, and tools/feed_nccl_selftest.py runs the feed over the real RCCL backend)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


COMPACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "parity_spot", "full_record"}
ROOFLINE_KEYS = {"bound", "achieved", "peak", "unit", "frac", "frac_algorithmic_shared", "traffic", "kernel", "kernel_ms", "units_per_launch"}


def _run(extra, env_extra=None, timeout=300, variants=False, compact=False):
    """-> the FULL record (the file `--full-json` names); compact=True: (compact stdout line as a dict, its length in bytes, full record).
    stdout carries exactly ONE line, the compact one, at most 4096 bytes (the driver parses an 8 KB tail of stdout: a 21.8 KB line cost
    round 5 its record)."""
    import tempfile

    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as td:
        fp = os.path.join(td, "full.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-test", "--steps", "2", "--warmup", "1", "--full-json", fp,
                            "--no-cpu-baseline"] + ([] if variants else ["--no-variants"]) + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        every = [l for l in r.stdout.splitlines() if l.strip()]
        lines = [l for l in every if l.startswith("{")]  # (gloo's transport prints "[Gloo] Rank ..." lines of its own on stdout)
        assert len(lines) == 1 and every[-1] == lines[0], r.stdout[-2000:]  # ONE JSON line, from rank 0, and it is the LAST line of stdout
        assert len(lines[0].encode()) <= 4096, len(lines[0])
        c = json.loads(lines[0])
        assert COMPACT_KEYS <= set(c) and ROOFLINE_KEYS <= set(c["roofline"]), sorted(c)
        assert {"workload", "clients_total", "blocks_per_call", "rccl_ranks", "parallelism"} <= set(c["config"])
        full = json.load(open(fp))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):  # the compact line is a projection of the record
        assert c[k] == full[k], k
    return (c, len(lines[0].encode()), full) if compact else full


def test_self_launch_two_ranks_strong_scaling():
    j = _run(["--gpus", "2", "--scaling", "strong"])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1
    assert j["scaling"] == "strong" and j["config"]["clients_total"] == 1024  # BASELINE configs[3]: 1024 clients in total
    assert j["config"]["rccl_ranks"] == 2
    assert "512 on this GPU" in j["config"]["workload"]
    assert j["data"] == "cpu-plumbing-test" and j["value"] > 0
    blocks_per_step = int(j["config"]["step"].split()[0])
    assert abs(j["ms_per_step"] * j["steps"] / 1e3 - j["config"]["us_per_block"] * blocks_per_step * j["steps"] / 1e6) < 1e-3
    # the timed region is repeated; the line reports the median repeat and shows all of them
    assert j["repeats"]["n"] == 3 and j["repeats"]["median"] == j["ms_per_step"] and len(j["repeats"]["values"]) == 3
    assert j["roofline"]["frac"] <= 1.0 and "model_frac" in j["roofline"]["algorithmic"]
    assert set(j["expected_scaling"]["strong"]["Msamples_per_s"]) == {"1", "2", "4", "8"}


def test_self_launch_weak_scaling_and_single_rank():
    c, nbytes, j = _run(["--gpus", "2", "--clients", "64"], compact=True)  # (weak scaling is the default: 64 clients per GPU)
    assert j["scaling"] == "weak" and j["config"]["clients_total"] == 128
    # the compact line of an N > 1 run: rank 0 prints it, config.rccl_ranks = N (what a first SCALE run parses)
    assert c["n_gpus"] == 2 and c["config"]["rccl_ranks"] == 2 and c["config"]["clients_total"] == 128 and c["config"]["clients_per_gpu"] == 64
    assert "expected_scaling" not in c and "variants" not in c and nbytes <= 4096
    j1 = _run(["--gpus", "1", "--clients", "64"])
    assert j1["n_gpus"] == 1 and j1["config"]["clients_total"] == 64 and j1["config"]["parallelism"] == "single GPU"


def test_compact_line_of_a_fully_populated_record_stays_under_4096_bytes():
    """compact_line() on a record shaped like a GPU run's (every variant present, long prose everywhere): <= 4096 bytes, nine `configs`
    entries of {value, us_per_block, parity_ok}, roofline + cpu_baseline + parity_spot carried, no prose carried along."""
    sys.path.insert(0, ROOT)
    import bench

    prose = "x" * 900
    spot = {"clients": 4096, "clients_failing": 0, "max_rel": 1.2345678e-06, "ok": True, "how": prose, "seconds": 12.5}
    var = lambda v: {"value": v, "us_per_block": 25.123456, "ms_per_step": 8.1, "plan": prose, "note": prose, "parity_spot": spot,
                     "roofline": {"frac": 0.5612, "frac_algorithmic_shared": 0.1534, "per_kernel": {"k": prose}}}  # noqa: E731
    full = {"metric": "input IQ Msamples/s processed (all clients), 2.016 Msps->48 kHz xlating FIR", "value": 5375384.0, "unit": "Msamples/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 47.9404, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            "dtype_note": prose, "plan": prose, "expected_scaling": {"why": prose}, "device": prose,
            "config": {"workload": prose, "clients_total": 1024, "clients_per_gpu": 1024, "ntaps": 505, "mode": "optimized", "blocks_per_call": 8,
                       "us_per_block": 24.97, "rccl_ranks": 1, "feed": prose, "mix_products": "2xf16 split operands on the matrix cores, f32 accumulate"},
            "roofline": {"bound": "hbm", "achieved": 3801.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.4752, "frac_algorithmic_shared": 0.1269,
                         "traffic": 774000000, "traffic_source": "measured in this run", "kernel_ms": 0.2035, "units_per_launch": 1073741824,
                         "call_period_ms": 0.2035, "step_bound": "nco_recurrence", "chain_ms_per_call": 0.201, "frac_of_recurrence_floor": 0.9877,
                         "kernel_short": "xlp_forward+xlp_mix_mfma+xlp_inverse32 (the 3 launches of one call)", "counter_scope": prose, "frac_is": prose,
                         "per_kernel": {k: {"ms": 0.10291234, "frac_hbm": 0.5406, "binding": prose} for k in
                                        ("xlp_forward_kernel", "xlp_mix_mfma_kernel", "xlp_inverse32_kernel", "xl_nco_chain_kernel")}},
            "parity_spot": spot, "native": var(1184327.7),
            "cpu_baseline": {"value": 3484.1, "unit": "Msamples/s", "cores": 128, "threads": 256, "kind": "reference", "single_thread_value": 730.2,
                             "sample": prose, "sample_short": "reference AVX2 -O3 -ffast-math build, 505 taps D=42, one filter per thread: 9000 calls / 256 threads / 8.4 s"},
            "variants": {"one block per call (the reference's call granularity)": var(3.3e6), "config 3: 64 clients at mixed 48 / 96 kHz": var(176000.0),
                         "config 2: one client, drop-in process_optimized_cu8_cf32": var(2900.0), "2048 clients on this GPU (kernel-bound regime)": var(6.4e6),
                         "4096 clients on this GPU (kernel-bound regime)": var(6.85e6), "config 5: cf32 10 Msps, D=100, 257 taps, 1024 clients": var(6.5e6),
                         "polyphase, float32 matrix-core mix (all-float32 products)": var(4.53e6), "host-delivered outputs (process_host + fetch per call)": var(278000.0),
                         "inverse launch A/B in this process": {"2048 clients": {"runs": [prose]}}, "lpf_cutoff_rate=1 (101 taps)": var(5.4e6)},
            "full_record": "profiles/bench_last_full.json"}
    line = bench.compact_line(full)
    assert len(line.encode()) <= 4096, len(line)
    c = json.loads(line)
    assert COMPACT_KEYS <= set(c) and ROOFLINE_KEYS <= set(c["roofline"]) and "xxxx" not in line
    assert set(c["configs"]) == {"one_block_per_call", "config3_64_mixed_clients", "config2_dropin_single_filter", "clients_2048", "clients_4096",
                                 "config5_cf32_10msps_1024_clients", "all_f32", "native", "host_delivered"}
    assert all(e["parity_ok"] is True and e["value"] > 0 and len(json.dumps(e)) <= 150 for e in c["configs"].values())
    assert c["cpu_baseline"]["cores"] == 128 and c["cpu_baseline"]["kind"] == "reference" and c["parity_spot"]["clients_failing"] == 0
    assert c["roofline"]["step_bound"] == "nco_recurrence" and c["roofline"]["frac_of_recurrence_floor"] == 0.9877
    assert c["roofline"]["per_kernel"]["xlp_inverse32"] == {"ms": 0.1029, "frac_hbm": 0.5406}


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-test", "--gpus", "2"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_two_ranks_line_carries_strong_and_weak_fully_populated():
    """N > 1: ONE line with both ways to use the GPUs -- the headline's scaling mode and the other one -- each with its value, step
    time and every rank's wall time of every timed repeat (the value uses the max over ranks)."""
    j = _run(["--gpus", "2", "--clients", "64"], variants=True)
    mg = j["multi_gpu"]
    assert set(mg) >= {"strong", "weak"} and mg["weak"]["value"] == j["value"] and mg["weak"]["clients_total"] == 128
    assert mg["strong"]["value"] > 0 and mg["strong"]["ms_per_step"] > 0
    for mode in ("strong", "weak"):
        prs = mg[mode]["per_rank_seconds"]
        assert len(prs) >= 1 and all(len(r) == 2 and min(r) > 0 for r in prs), prs


def test_a_rank_without_engine_stops_every_rank_within_the_deadline():
    """One rank cannot create its engine: every rank learns it (an all-reduce right after the create) and exits with an error;
    nobody is left waiting in a barrier or a broadcast."""
    env = dict(os.environ, XL_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-test", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-variants", "--no-cpu-baseline"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "could not be created on every rank" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no measurement line


def test_counter_pass_rows_are_split_per_workload(tmp_path):
    """bench.py replays several workloads in ONE rocprofv3 process per counter and splits the rows by the stand-alone phase-table
    launches only a fresh engine makes (two per engine: its first and its second call): a synthetic counter file with three
    workloads -- template arguments in the kernel names, rows out of dispatch order, foreign kernels in between -- comes apart into
    the three workloads, each with its steady-state kernels and none of the other workloads' rows."""
    import csv
    import random

    sys.path.insert(0, ROOT)
    import bench

    rows, did = [], 0

    def emit(name, value):
        nonlocal did
        did += 1
        rows.append({"Dispatch_Id": did, "Kernel_Name": name, "Counter_Name": "FETCH_SIZE", "Counter_Value": value})

    shapes = [("void xlp_mix_mfma_kernel<6>(XlpArgs)", "xlp_inverse8_kernel(XlpArgs)", 100.0),
              ("void xlp_mix_mfma_kernel<6>(XlpArgs)", "void xlp_inverse_kernel<128, XlpPosSwz>(XlpArgs)", 200.0),
              ("void xlp_mix_f32_kernel<13>(XlpArgs)", "xlp_inverse8_kernel(XlpArgs)", 300.0)]
    for mix, inv, v in shapes:
        emit("xl_nco_table_kernel(XlNcoClient const*, unsigned int)", 1.0)
        emit("void xl_fir_kernel<10, 1, true>(XlFirArgs)", 5.0)  # the fresh engine's first call: clients still inside their zero history
        emit("__amd_rocclr_copyBuffer", 7.0)
        emit("xl_nco_table_kernel(XlNcoClient const*, unsigned int)", 1.0)
        for c in range(16):
            emit("void xlp_forward_kernel<128>(XlpArgs)", v)
            emit(mix, 2 * v)
            emit(inv, 3 * v)
            if c % 4 == 0:
                emit("xl_nco_chain_kernel(XlNcoClient const*)", 4 * v)
    random.Random(3).shuffle(rows)
    path = tmp_path / "counter_collection.csv"
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    parsed = bench._pmc_rows(str(path), lambda row: float(row["Counter_Value"]))
    assert [r[0] for r in parsed] == sorted(r[0] for r in parsed) and all("rocclr" not in r[1] for r in parsed)
    segs = bench._split_workloads(parsed, 3)
    assert segs is not None and len(segs) == 3
    for (mix, inv, v), seg in zip(shapes, segs):
        assert len(seg["xlp_forward_kernel"]) == 16 and set(seg["xlp_forward_kernel"]) == {v}
        assert len(seg[mix.split("(")[0].replace("void ", "").split("<")[0]]) == 16
        assert len(seg["xl_nco_chain_kernel"]) == 4 and len(seg["xl_nco_table_kernel"]) == 2 and len(seg["xl_fir_kernel"]) == 1
    assert bench._split_workloads(parsed, 2) is None  # (a pass that does not show the workloads it was asked for is not used)
