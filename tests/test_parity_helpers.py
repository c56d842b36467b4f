"""points2surf_amd.parity: magnitude comparison at flipped signs, tie threshold (no GPU)."""
import numpy as np
import pytest

from points2surf_amd import parity


def test_compare_sdf_reports_flips_and_compares_their_magnitudes():
    ref = np.array([0.5, -0.25, 0.06695648, -0.01, 0.0], np.float32)
    sdf = np.array([0.5 + 1e-6, -0.25, -0.06695664, -0.01 - 2e-6, 0.0], np.float32)
    c = parity.compare_sdf(sdf, ref)
    assert list(c['flipped']) == [2]
    assert c['max_abs_dsdf'] < 3e-6                       # the flipped query counts with | |sdf| - |ref| |, not 0.13
    assert parity.compare_sdf(ref, ref)['flipped'].size == 0
    with pytest.raises(ValueError):
        parity.compare_sdf(sdf[:3], ref)


def test_tie_threshold():
    assert parity.not_ties([-4.0e-6, -2.2e-6]) == 0       # the two ties of the 512^3 grid (fp32 encoder)
    assert parity.not_ties([6e-5, -1e-6, -3e-3]) == 2
    assert parity.not_ties([]) == 0
    # the fp32 encoder has its own, tighter threshold than the split-precision modes
    assert parity.tie_logit(0) == parity.TIE_LOGIT_FP32 < parity.TIE_LOGIT_SPLIT == parity.tie_logit(4) == parity.tie_logit(3)
    assert parity.not_ties([1.2e-5]) == 1 and parity.not_ties([1.2e-5], encoder_bf16=4) == 0
    assert parity.not_ties([2.4e-7, -1.4e-7]) == 0        # the one tie of the 256^3 three-cloud dataset


def test_two_logit_rule():
    """parity.is_tie: the device's AND the CPU restatement's sign logit within the mode's threshold"""
    assert parity.is_tie(-6.5e-6, -7.4e-6)                # query 46,971 of the 512^3 grid: the reference against itself
    assert not parity.is_tie(2e-6, 3e-5) and not parity.is_tie(3e-5, 2e-6)
    assert not parity.is_tie(1.5e-5, 1.5e-5) and parity.is_tie(1.5e-5, -1.5e-5, encoder_bf16=4)
