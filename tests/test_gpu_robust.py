"""GPU: error paths of the per-shape pipeline (VERDICT r1 'robustness' item 10): an injected failing chunk leaves no
leaked buffers, drains both streams, and the next call on the same handles gives the exact result."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 40938661


def test_failing_chunk_is_reported_and_nothing_leaks(fixture_cloud):
    import torch
    from points2surf_amd import engine, synth, _lib
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    rng = engine.Rng(SEED)                       # one generator handle (its session buffer is allocated once)
    mt0, pos0 = rng.get_state()
    ref, _ = engine.infer_shape(model, cloud, rng, 32, 3, chunk=512)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for trial in range(6):
        model.debug_fault_chunk(trial % 3 + 1)
        try:
            engine.infer_shape(model, cloud, rng, 32, 3, chunk=512)
            raise AssertionError('the injected fault was not reported')
        except _lib.P2SError as e:
            assert e.code == -2 and 'injected fault' in str(e)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), (free0, free1)             # pipeline buffers are model-owned: no per-call growth
    rng.set_state(mt0, pos0)
    again, _ = engine.infer_shape(model, cloud, rng, 32, 3, chunk=512)
    torch.cuda.synchronize()
    assert torch.equal(ref, again)


def test_pipeline_buffers_are_reused_across_shapes_and_chunk_sizes(fixture_cloud):
    import torch
    from points2surf_amd import engine, synth
    w, cfg = synth.make_weights('p2s_max')
    model = engine.Model(w, cfg)
    cloud = engine.Cloud(fixture_cloud)
    small = engine.Cloud(fixture_cloud[::3])
    rng = engine.Rng(SEED)
    mt0, pos0 = rng.get_state()
    a, _ = engine.infer_shape(model, cloud, rng, 32, 3, chunk=1000)
    engine.infer_shape(model, small, rng, 24, 3, chunk=300)       # smaller chunk: same buffers
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        engine.infer_shape(model, small, rng, 24, 3, chunk=300)
        rng.set_state(mt0, pos0)
        b, _ = engine.infer_shape(model, cloud, rng, 32, 3, chunk=1000)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert free0 - torch.cuda.mem_get_info()[0] < (8 << 20)
    assert torch.equal(a, b)
