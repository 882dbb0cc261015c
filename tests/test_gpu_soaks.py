"""GPU: the randomised soaks of tools/ (round 6) for a few seconds each, so that they stay runnable and every driver run repeats a slice
of them: random moduli / ring dimensions / batches, EVERY instance of every launch compared with the oracle on the device. The long runs
that found the inverse-after-inverse LDS race and the 62-bit dyadic product are profiles/r06_inverse_after_inverse_race.txt,
r06_soaks_round3_and_dyadic.txt and r06_final_soak.txt."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("tool,args,marker", [("soak_ks_random.py", ["10", "17"], "mismatches: 0"),
                                              ("soak_ntt_random.py", ["8", "18", "3"], "mismatching launches: 0"),
                                              ("soak_dyadic_random.py", ["6", "19"], "mismatches: 0")])
def test_soak_slice(tool, args, marker):
    out = subprocess.run([sys.executable, str(ROOT / "tools" / tool)] + args, capture_output=True, text=True, timeout=600)
    print(out.stdout[-1500:], out.stderr[-1500:])
    assert out.returncode == 0 and marker in out.stdout
