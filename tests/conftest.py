import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_keep = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box: a SIGABRT handler that leaves the aborting thread's native call stack and the tail of the captured stderr in a
    # file (tools/r06/abort_shim.c).  Round 5 saw two of six whole-suite runs end with "Fatal Python error: Aborted" and nothing
    # else -- pytest's capture owns file descriptor 2, and whatever the HIP runtime or glibc said before abort() went with it.
    # Ten runs of the same suite in round 6 (sweeps in-process, as then) did not abort; if one ever does again, the cause is named
    # in gpurun_out/abort_<pid>.log.  Loaded before faulthandler is (re-)enabled, so that its Python traceback lands there too.
    shim = os.path.join(ROOT, "tools", "r06", "libabort_shim.so")
    if os.path.exists(shim) and os.path.exists("/dev/kfd") and not os.environ.get("SRLA_NO_ABORT_SHIM"):
        import ctypes
        import faulthandler
        out = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(out, exist_ok=True)
            os.environ.setdefault("SRLA_ABORT_LOG", os.path.join(out, "abort_%d.log" % os.getpid()))
            _keep.append(ctypes.CDLL(shim))
            f = open(os.environ["SRLA_ABORT_LOG"] + ".py", "w")
            _keep.append(f)
            faulthandler.enable(file=f, all_threads=True)
        except OSError:
            pass


def pytest_sessionfinish(session, exitstatus):
    # what a long-lived test process holds at its end (threads, descriptors, memory): the numbers a leak would show in
    try:
        st = dict(l.split(":", 1) for l in open("/proc/self/status").read().splitlines() if ":" in l)
        line = "pid %d: Threads %s, VmRSS %s, VmSize %s, fds %d, exit status %s\n" % (
            os.getpid(), st.get("Threads", "?").strip(), st.get("VmRSS", "?").strip(), st.get("VmSize", "?").strip(),
            len(os.listdir("/proc/self/fd")), exitstatus)
        if os.path.exists("/dev/kfd"):
            with open(os.path.join(ROOT, "gpurun_out", "test_process_at_exit.txt"), "a") as f:
                f.write(line)
    except OSError:
        pass
    for f in _keep:
        if hasattr(f, "name") and hasattr(f, "tell"):
            try:
                empty = f.tell() == 0
                f.close()
                if empty:
                    os.unlink(f.name)
            except OSError:
                pass


@pytest.fixture(scope="session")
def product():
    """The MI355X library through the reference's own C API binding.  No fallback: a missing
    library is a failure, not a skip."""
    import helpers
    from srla_amd import capi
    assert os.path.exists(helpers.PRODUCT_SO), "srla_amd/libsrla_mi355x.so is not built (run __graft_entry__.build())"
    return capi.EncoderLib(helpers.PRODUCT_SO)
