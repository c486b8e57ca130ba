import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "web-splat_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "experimental: exercises a measured-and-lost variant that only the experimental build "
                                       "(make experimental, WEBSPLAT_LIB=.../lib_exp/libwebsplat_hip.so) contains")


from variants import lib_is_experimental  # noqa: E402  (tests/variants.py: the explicit marking of the variant tests)


def pytest_collection_modifyitems(config, items):
    marked = [it for it in items if it.get_closest_marker("experimental")]
    if not marked:
        return
    if lib_is_experimental():
        return
    skip = pytest.mark.skip(reason="measured-and-lost variant: experimental build only (WEBSPLAT_LIB=.../lib_exp/libwebsplat_hip.so)")
    for it in marked:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def ws():
    import websplat
    return websplat


@pytest.fixture(scope="session")
def ctx(ws):
    c = ws.Context(0)
    yield c
    c.close()


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: how many depth keys of the compared K1 frames were not bit-identical to the oracle's / the reference
    shader's (tests/test_gpu_preprocess.py collects one entry per frame) -> gpurun_out/k1_key_report.json."""
    mod = sys.modules.get("test_gpu_preprocess")
    rep = getattr(mod, "KEY_REPORT", None) if mod else None
    if rep:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        tot = {"frames": len(rep), "keys": sum(r["keys"] for r in rep), "differ": sum(r["differ"] for r in rep),
               "max": max(r["max"] for r in rep)}
        with open(os.path.join(out, "k1_key_report.json"), "w") as f:
            json.dump({"total": tot, "frames": rep}, f, indent=1)
