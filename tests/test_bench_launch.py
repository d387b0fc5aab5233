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


def _run(extra, env_extra=None, timeout=300, variants=False):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-test", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"] + ([] if variants else ["--no-variants"]) + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


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
    j = _run(["--gpus", "2", "--clients", "64"])  # (weak scaling is the default: 64 clients per GPU)
    assert j["scaling"] == "weak" and j["config"]["clients_total"] == 128
    j1 = _run(["--gpus", "1", "--clients", "64"])
    assert j1["n_gpus"] == 1 and j1["config"]["clients_total"] == 64 and j1["config"]["parallelism"] == "single GPU"


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
