"""A short run of the differential fuzzer (scripts/fuzz_parity.py) inside the
GPU suite: fixed seeds, every call mode drawn at random -- the long campaigns
are recorded in profiles/sessions.md (r4_session4x / r4_session5x)."""
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (seeds 23 / 51 in these modes are where the invalid launches were first seen;
# the random stream has changed with the fuzzer's menus since, and the defects
# have tests of their own: test_abi_and_host_logic.py
# ::test_a_launch_is_sized_for_two_maxima_and_must_still_fit_the_lds)
@pytest.mark.parametrize('seed, modes, rounds', [
    (7, None, 10),
    (23, 'retheta', 8),
    (51, 'bulk', 10),
    (4, 'sym,lmin,nodal', 4),  # (tiny graphs among the sizes; the float-rounded
    #                            one-row systems have a test of their own)
    # round 5: the pair-list API; Normalize(DotProduct()) / Convolution on
    # nodes and edges of sparse and dense graphs; spatial graphs of 65-300
    # nodes (16-wave on-the-fly, streamed and general solvers); two ranks
    # through the pair-sharded path
    (11, 'pairlist', 6),
    (12, 'features', 6),
    (13, 'spatial', 2),
    (14, 'sharded', 2),
    # round 6: the dense-tile solver on the matrix cores (mgk_mfma.h)
    (15, 'mfma', 5),
])
def test_fuzzer_rounds(seed, modes, rounds):
    cmd = [sys.executable, os.path.join(ROOT, 'scripts', 'fuzz_parity.py'),
           str(rounds), f'--seed={seed}'] \
        + ([f'--modes={modes}'] if modes else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500,
                       cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and 'fuzz ok' in r.stdout, tail
