"""bench.py's helpers that need no GPU: the counter passes it runs as child processes must come back with a reason, not an
exception, on a host where they cannot measure (no GPU, or no rocprofv3), so that the bench line falls back to the
committed figures and says why."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_counter_passes_report_a_reason_where_they_cannot_measure():
    import torch

    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the passes would measure")
    import bench

    got, why = bench.counter_pass(["--spp", "1", "--width", "16", "--height", "8"], ["FETCH_SIZE"], timeout_s=120)
    assert got is None and isinstance(why, str) and why
    total, text = bench.measured_traffic(["--spp", "1", "--width", "16", "--height", "8"])
    assert total is None and text
    live, why = bench.measured_executed(["--spp", "1", "--width", "16", "--height", "8"])
    assert live is None and why


def test_recorded_summaries_are_readable():
    import bench

    assert bench.recorded("hbm_traffic.json", "c2/1920x1080")["fixed"] > 0
    entry = bench.recorded("pmc_summary.json", "c2/LDS2")
    assert 0.0 < entry["lane_utilisation"] <= 1.0 and entry["source"].startswith("profiles/")
    assert os.path.exists(os.path.join(ROOT, entry["source"]))
    assert bench.recorded("pmc_summary.json", "no such key") is None
