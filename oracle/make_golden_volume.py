"""TEST INFRASTRUCTURE ONLY -- golden volumes for the "next" row f-1 from the UNMODIFIED reference
(source/sdf.py add_samples_to_volume + propagate_sign), fed with the committed golden SDF of the fixture
shape.  Run in the build container:  python -m oracle.make_golden_volume"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle import ref_shims  # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def main():
    ref_shims.install()
    from source import sdf as ref_sdf
    q = np.load(os.path.join(GOLDEN, 'query_grid_32_3.npy'))
    out = {}
    for model in ('p2s_max', 'p2s_vanilla'):
        sdf = np.load(os.path.join(GOLDEN, 'ref_%s_grid32.npz' % model))['sdf_full']
        for sigma, thr in ((5, 13), (3, 5), (4, 9.5), (2, 3), (6, 40)):   # even sigma: scipy's convolve origin
            vol = np.zeros((32, 32, 32))
            vol = ref_sdf.add_samples_to_volume(vol, q, sdf)
            vol = ref_sdf.propagate_sign(vol, sigma, thr)
            vol[vol < -1.0] = -1.0
            vol[vol > 1.0] = 1.0
            assert np.array_equal(vol.astype(np.float32).astype(np.float64), vol)   # float32 is lossless here
            out['%s_s%d_t%g' % (model, sigma, thr)] = vol.astype(np.float32)
            print(model, sigma, thr, 'neg/zero/pos', (vol < 0).sum(), (vol == 0).sum(), (vol > 0).sum())
    np.savez_compressed(os.path.join(GOLDEN, 'ref_volume_grid32.npz'), **out)


if __name__ == '__main__':
    main()
