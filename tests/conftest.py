import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the CPU-side libraries exist (oracle, VTK writer, C-ABI .so)."""
    import __graft_entry__ as g

    g.build(only_missing=True)


@pytest.fixture(autouse=True)
def _schedule_hazard_check(request, tmp_path, monkeypatch):
    """Every GPU test doubles as a schedule test: libsfgpu.so traces the launches, exchanges and stream-ordering calls
    of the contexts the test creates (SF_TRACE_SCHEDULE), and tests/schedule_check.py must find no unordered
    conflicting accesses afterwards — deterministic, whatever the timing of this particular run was."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("SF_TRACE_SCHEDULE"):
        yield
        return
    path = tmp_path / "schedule_trace.jsonl"
    monkeypatch.setenv("SF_TRACE_SCHEDULE", str(path))
    yield
    if path.exists() and path.stat().st_size < (64 << 20):
        import schedule_check as SC

        ctxs, bad = SC.check_file(str(path))
        assert not bad, "launch schedule hazard:\n" + SC.describe(bad)
