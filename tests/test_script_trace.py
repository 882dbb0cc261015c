"""CPU: the reference's runner scripts (tests/micro_*.sh, benchmark/micro_*.sh) are executed UNMODIFIED, as scripts, against recording
stubs (tests/ref_harness/trace_scripts.py); the GPU suite replays what they executed. Only where the reference tree exists."""
import json
import sys
from pathlib import Path

import pytest

HARNESS = Path(__file__).resolve().parent / "ref_harness"
REF = Path("/root/reference")


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present on this box")
def test_reference_scripts_run_unmodified_and_trace(tmp_path):
    sys.path.insert(0, str(HARNESS))
    import trace_scripts
    out = tmp_path / "trace.json"
    assert trace_scripts.trace(REF, out) == 0
    t = json.loads(out.read_text())
    # every runner script of the reference was run, in its own directory's terms, and called at least one known binary
    assert sorted(t["scripts"]) == sorted(str(p.relative_to(REF)) for p in list(REF.glob("tests/micro_*.sh")) + list(REF.glob("benchmark/micro_*.sh")))
    by = {}
    for e in t["invocations"]:
        by.setdefault(e["script"], []).append(e)
        assert e["env"].get("FPGA_BITSTREAM", "").startswith("/nonexistent-bitstreams/")      # bitstream_dir.sh was sourced
        assert "FPGA_KERNEL" in e["env"]
    # spot checks against the scripts' own command lines (micro_keyswitch.sh:20-34, benchmark/micro_keyswitch.sh)
    ks = [(e["env"].get("N"), e["env"].get("BATCH_SIZE_KEYSWITCH")) for e in by["tests/micro_keyswitch.sh"]]
    assert ks == [(None, None), ("16384", "1"), ("16384", "2"), ("8192", "1"), ("8192", "1")]
    bk = [(e["env"]["ITER"], e["env"]["BATCH_SIZE_KEYSWITCH"]) for e in by["benchmark/micro_keyswitch.sh"]]
    assert bk == [("256", "1"), ("256", "16"), ("256", "128")]


def test_committed_build_trace_matches_a_fresh_one_when_both_exist(tmp_path):
    built = HARNESS / "_build" / "script_trace.json"
    if not (REF.exists() and built.exists()):
        pytest.skip("needs the reference tree and a built harness")
    sys.path.insert(0, str(HARNESS))
    import trace_scripts
    out = tmp_path / "trace.json"
    assert trace_scripts.trace(REF, out) == 0
    assert json.loads(out.read_text())["invocations"] == json.loads(built.read_text())["invocations"]
