"""The block-wise resolution of the legacy shuffle (tests/radius_model.py) against numpy itself: permutation and the
number of 32-bit words consumed, for counts around every power of two and the sizes fixed-radius patches have."""
import numpy as np
import pytest

from radius_model import permutation, raw_words, walk_blocks


@pytest.mark.parametrize('lanes', [64, 256])
@pytest.mark.parametrize('seed', [0, 1, 40938661])
def test_blockwise_shuffle_is_numpys_permutation(seed, lanes):
    rs = np.random.RandomState(seed)
    rs.randint(0, 1000, size=seed % 97)                    # some position inside a block
    for n in [2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 301, 512, 513, 813, 1024, 1025, 3643, 5903]:
        words = raw_words(rs, 3 * n + 1024)
        got, used = permutation(words, 0, n, lanes=lanes)
        ref = rs.permutation(n)
        assert np.array_equal(got, ref), n
        # consumption: the generator now stands `used` words further
        assert np.array_equal(raw_words(rs, 4), words[used:used + 4]), n


def test_fixed_point_needs_few_evaluations():
    rs = np.random.RandomState(5)
    words = raw_words(rs, 20000)
    _, _, blocks, evals = walk_blocks(words, 0, 5903)
    assert blocks <= 140 and evals / blocks < 1.5
