import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

# the child-process legs of a GPU session (tests/preflight.py): {"proc": Popen, "dir": path} once started
PREFLIGHT = {}


def _gpu_run_selected(config):
    """-m gpu (or no marker filter at all) on a box that has a device.  torch.cuda.device_count() does not
    initialise the GPU on this image; torch.cuda.is_available() would."""
    expr = config.getoption("-m") or ""
    if "not gpu" in expr:
        return False
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:  # pragma: no cover
        return False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A test that hangs (a rendezvous, a child process, a wedged launch) must fail, not hold the session until somebody's outer
    # limit kills it with nothing reported: with pytest-timeout present (it is in this image) and no --timeout given, every test
    # gets 900 s -- the slowest one takes ~3 min on a fresh box (hipcc builds of the deliberately broken kernels).
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900.0
    # MF_TEST_WGRAD=f32|bf16x3: run the whole session with that arithmetic of the weight-gradient contractions (the
    # gradient bars must hold in both; the package default is what a plain run tests)
    if os.environ.get("MF_TEST_TRAIN_FWD"):
        from moco_flow_amd import rendering as _Rn
        _Rn.set_train_forward_precision(os.environ["MF_TEST_TRAIN_FWD"])
    if os.environ.get("MF_TEST_DX"):
        from moco_flow_amd import autograd as _A2
        _A2.set_dx_precision(os.environ["MF_TEST_DX"])
    if os.environ.get("MF_TEST_WGRAD"):
        from moco_flow_amd import autograd as _A
        _A.set_wgrad_precision(os.environ["MF_TEST_WGRAD"])
    # Start the child-process legs NOW (tests/preflight.py: the 1-rank RCCL child, then bench.py --gpus 2), while this
    # process has not touched the GPU: a process that has initialised the GPU must not start another GPU program on this
    # pool, and every later point of a -m gpu session is behind such a call.
    if _gpu_run_selected(config) and not os.environ.get("MF_NO_PREFLIGHT") and not hasattr(config, "workerinput"):
        out = tempfile.mkdtemp(prefix="mf_preflight_")
        PREFLIGHT["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "preflight.py"), out], cwd=ROOT)
        PREFLIGHT["dir"] = out


def pytest_unconfigure(config):
    p = PREFLIGHT.get("proc")
    if p is not None and p.poll() is None:       # the exact process we started (its children carry their own timeouts)
        p.kill()


def pytest_collection_modifyitems(config, items):
    """GPU tests are selected with ``-m gpu``; anywhere else they are skipped when no
    device is present (so a plain ``pytest tests`` in the build container stays green)."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def preflight():
    """{"status": {"rccl": rc, "bench2": rc}, "dir": logs} of tests/preflight.py, or {} when it was not started.  Autouse:
    the session's first test waits for the children to finish, so they never share the GPU with a timing- or
    bit-sensitive test (their own timeouts bound the wait)."""
    p = PREFLIGHT.get("proc")
    if p is None:
        return {}
    try:
        p.wait(timeout=1800)
    except subprocess.TimeoutExpired:
        p.kill()
    import json
    try:
        with open(os.path.join(PREFLIGHT["dir"], "status.json")) as fh:
            status = json.load(fh)
    except OSError:
        status = {}
    return {"status": status, "dir": PREFLIGHT["dir"]}


@pytest.fixture(params=["f32", "bf16x3"])
def wgrad(request):
    """Arithmetic of the backward's matrix work -- the weight-gradient contractions (autograd.set_wgrad_precision) and the
    NeRF's input-gradient chain (set_dx_precision) -- for the duration of one test: the exact-fp32 MFMA kernels and the
    three-product bf16 ones must both hold the gradient bars."""
    from moco_flow_amd import autograd as A
    old = (A.WGRAD_PRECISION, A.DX_PRECISION)
    A.set_wgrad_precision(request.param)
    A.set_dx_precision(request.param)
    yield request.param
    A.set_wgrad_precision(old[0])
    A.set_dx_precision(old[1])
