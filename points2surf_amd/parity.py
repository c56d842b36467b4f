"""Comparing an SDF array with a reference one (bench.py's self-check and the full-grid parity tests).

The reference decides the sign as ``sign logit >= 0`` (source/sdf_nn.py:16-21): a query whose sign logit lies within
the fp32 noise of zero (|logit| of a few 1e-6 at a logit accuracy of ~1.5e-5; about 1 query in 400,000) has no
reproducible sign -- the reference itself answers differently for it depending on its batch composition.  Such
queries are compared by magnitude and reported separately; whether a flip really is such a tie is decided by the
caller from the device's own sign logit (engine.query_logits)."""
import numpy as np

TIE_LOGIT = 2e-5       # |sign logit| below which a flipped sign counts as an fp32 tie (largest tie observed: 8e-6;
                       # logit accuracy of the fp32 path ~1.5e-5)


def compare_sdf(sdf, ref):
    """-> dict(max_abs_dsdf, flipped): max |sdf - ref| with the MAGNITUDES compared where the signs differ, and the
    indices of those queries"""
    sdf, ref = np.asarray(sdf), np.asarray(ref)
    if sdf.shape != ref.shape:
        raise ValueError('shape %s vs reference %s' % (sdf.shape, ref.shape))
    flipped = np.nonzero(np.sign(sdf) != np.sign(ref))[0]
    d = np.abs(sdf - ref)
    d[flipped] = np.abs(np.abs(sdf[flipped]) - np.abs(ref[flipped]))
    return {'max_abs_dsdf': float(d.max()) if d.size else 0.0, 'flipped': flipped}


def not_ties(sign_logits):
    """how many of the flipped queries are NOT fp32 ties, given the device's sign logits for them"""
    return int(sum(abs(float(x)) >= TIE_LOGIT for x in sign_logits))
