"""Build helper: compiles csrc/ for gfx950 into lib/libxlating_hip.so (in-tree, travels with gpurun)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def library_path():
    # XL_LIBRARY_PATH: test infrastructure only (tools/sanitize.sh points the host-code tests at an instrumented build,
    # tools/experiments/build_variant.sh at a variant of one kernel file) -- honoured only next to XL_TESTING=1, so that a stray
    # environment variable cannot redirect a production process to another shared object
    if os.environ.get("XL_TESTING") == "1" and os.environ.get("XL_LIBRARY_PATH"):
        return os.environ["XL_LIBRARY_PATH"]
    return os.path.join(HERE, "lib", "libxlating_hip.so")


def build_library(verbose=False):
    env = dict(os.environ)
    env.setdefault("PATH", "")
    if "/opt/rocm/bin" not in env["PATH"]:
        env["PATH"] = "/opt/rocm/bin:" + env["PATH"]
    r = subprocess.run(["make", "-C", os.path.join(HERE, "csrc")], env=env, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libxlating_hip.so failed")
    return library_path()
