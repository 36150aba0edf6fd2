"""pytest plugin (loaded through pytest.ini `-p tests._autoparallel`): on a box WITHOUT a GPU the suite is the CPU suite -- the
kernels run on the fiber-based HIP emulator, which is slow and single-threaded per test -- so spread the test files over the host
cores with pytest-xdist.  On a GPU box (the `-m gpu` run) nothing changes: one process, tests in order."""
import os


def pytest_load_initial_conftests(early_config, parser, args):
    if os.environ.get("Y5_TEST_WORKERS") == "0" or any(a == "-n" or a.startswith("-n") or a.startswith("--numprocesses") or a == "-p" and False for a in args):
        return
    try:
        import xdist  # noqa: F401
        import torch

        if torch.cuda.is_available():
            return
    except Exception:
        return
    n = os.environ.get("Y5_TEST_WORKERS") or str(max(1, min(8, (os.cpu_count() or 2) - 1)))
    # torch's intra-op pool would otherwise start one thread per core in EVERY worker (8 x 7 threads on 8 cores)
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, "2")
    args[:] = ["-n", n, "--dist", "worksteal"] + list(args)
