"""Boundary (B1/B2): the drop-in modules expose the reference's API surface.  CPU only."""
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, 'points2surf_amd', 'dropin')


@pytest.fixture()
def dropin_source():
    saved = {k: v for k, v in sys.modules.items() if k == 'source' or k.startswith('source.')}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, DROPIN)
    try:
        import source.points_to_surf_eval as ev
        import source.points_to_surf_model as mo
        yield ev, mo
    finally:
        sys.path.remove(DROPIN)
        for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_state_dict_layout_matches_spec(dropin_source):
    from points2surf_amd import model_spec, synth
    ev, mo = dropin_source
    for name in ('p2s_max', 'p2s_vanilla', 'p2s_uniform', 'p2s_regression', 'p2s_shared_encoder'):
        c = model_spec.NAMED_MODELS[name]
        od, single = int(c.get('output_dim', 2)), bool(c.get('single_transformer', False))
        m = mo.PointsToSurfModel(net_size_max=1024, num_points=300, output_dim=od, use_point_stn=c['use_point_stn'],
                                 use_feat_stn=True, sym_op='max', use_query_point=True, sub_sample_size=1000,
                                 do_augmentation=False, single_transformer=int(single),
                                 shared_transformation=c['shared_transformation'])
        spec = model_spec.state_shapes(use_point_stn=c['use_point_stn'], shared_transformation=c['shared_transformation'],
                                       output_dim=od, single_transformer=single)
        sd = m.state_dict()
        assert set(sd.keys()) == set(spec.keys())
        for k, shape in spec.items():
            assert tuple(sd[k].shape) == tuple(shape), k
        # a reference-format checkpoint (DataParallel 'module.' prefix) loads strictly
        w, _ = synth.make_weights(name)
        torch.nn.DataParallel(m).load_state_dict(synth.to_torch_state_dict(w))
        if name in ('p2s_max', 'p2s_vanilla'):
            assert len(sd) == {'p2s_max': 174, 'p2s_vanilla': 211}[name]     # key counts of the reference (golden meta)
        if single:
            assert 'feat_local_global.stn1.fc3.weight' in sd and tuple(sd['fc1_local_global.weight'].shape) == (1024, 1024)
        if od == 1:
            assert tuple(sd['fc4.weight'].shape) == (1, 128)


def test_parse_arguments_defaults_and_quirks(dropin_source):
    ev, _ = dropin_source
    o = ev.parse_arguments([])
    assert o.dataset == 'testset.txt' and o.seed == 40938661 and o.models == 'p2s_vanilla'
    assert o.reconstruction is False and o.query_grid_resolution is None and o.batchSize == 0
    o = ev.parse_arguments(['--dataset', 'a/testset.txt', 'b/testset.txt', '--query_grid_resolution', '256',
                            '--epsilon', '3', '--certainty_threshold', '13', '--sigma', '5', '--workers', '7',
                            '--batchSize', '501', '--cache_capacity', '5', '--modelpostfix', '_model_249.pth',
                            '--models', 'p2s_max', '--indir', 'datasets', '--outdir', 'results', '--modeldir', 'models'])
    assert o.dataset == ['a/testset.txt', 'b/testset.txt'] and o.query_grid_resolution == 256 and o.sigma == 5


@pytest.mark.skipif(not os.path.isfile('/root/reference/source/points_to_surf_eval.py'), reason='reference not present')
def test_parse_arguments_identical_to_reference(dropin_source):
    ev, _ = dropin_source
    from oracle import ref_shims
    args = ['--query_grid_resolution', '128', '--epsilon', '3', '--models', 'p2s_max', '--dataset', 'x.txt']
    mine_default, mine = vars(ev.parse_arguments([])), vars(ev.parse_arguments(args))
    for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
        del sys.modules[k]
    sys.path.remove(DROPIN)
    try:
        with ref_shims.reference():
            from source import points_to_surf_eval as ref_ev
            assert vars(ref_ev.parse_arguments([])) == mine_default
            assert vars(ref_ev.parse_arguments(args)) == mine
    finally:
        sys.path.insert(0, DROPIN)


def test_errors_mirror_reference(dropin_source):
    ev, mo = dropin_source
    with pytest.raises(ValueError):
        mo.PointsToSurfModel(sym_op='median')
    o = ev.parse_arguments(['--gpu_idx', '-1'])
    o.reconstruction = True
    with pytest.raises(RuntimeError):
        ev.points_to_surf_eval(o)
    m = mo.PointsToSurfModel(net_size_max=1024, num_points=300, output_dim=2, use_point_stn=False,
                             sub_sample_size=1000).eval()
    with pytest.raises(RuntimeError):      # CPU tensors: loud failure, no fallback
        m({'patch_pts_ps': torch.zeros(1, 300, 3), 'pts_sub_sample_ms': torch.zeros(1, 1000, 3),
           'imp_surf_query_point_ms': torch.zeros(1, 3)})


@pytest.mark.skipif(not os.path.isfile('/root/reference/source/sdf.py'), reason='reference not present')
def test_sdf_bridge_reexports_reference_and_overrides_two_functions():
    """drop-in source.sdf: reference module re-exported, only the volume functions swapped (no GPU call here)"""
    import numpy as np
    from oracle import ref_shims
    saved = {k: v for k, v in sys.modules.items() if k == 'source' or k.startswith('source.')}
    for k in saved:
        del sys.modules[k]
    ref_shims._install_trimesh_stub()
    if not hasattr(np, 'int'):
        np.int = int
    sys.path.insert(0, ref_shims.REFERENCE_ROOT)
    sys.path.insert(0, DROPIN)
    try:
        import source.sdf as sdf
        assert sdf.__file__.startswith(DROPIN)
        ref = sys.modules['source._reference_sdf']
        assert ref.__file__.startswith(ref_shims.REFERENCE_ROOT)
        # re-exported reference functions (query grid helper is the reference's own object)
        assert sdf.get_voxel_centers_grid_smaller_pc is ref.get_voxel_centers_grid_smaller_pc
        # the consumer stage full_eval.py calls is the drop-in's own (device volume + iso-surface, serial driver: HIP
        # contexts do not survive the reference's fork pool)
        for name in ('implicit_surface_to_mesh', 'implicit_surface_to_mesh_file', 'implicit_surface_to_mesh_directory'):
            assert getattr(sdf, name) is not getattr(ref, name, None) and getattr(sdf, name).__module__ == 'source.sdf'
        assert ref.visualize_query_points is sdf.visualize_query_points and sdf.visualize_query_points.__module__ == 'source.sdf'
        # the two overridden names are patched into the reference module too
        assert ref.propagate_sign is sdf.propagate_sign and ref.add_samples_to_volume is sdf.add_samples_to_volume
        v = sdf.add_samples_to_volume(np.zeros((8, 8, 8)), np.zeros((1, 3), np.float32), np.ones(1, np.float32))
        assert v.shape == (8, 8, 8) and v._p2s_samples[1][0] == 1.0
        import torch
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError):       # no GPU -> loud failure, not the reference's CPU loop
                sdf.propagate_sign(v, 5, 13)
    finally:
        sys.path.remove(DROPIN)
        sys.path.remove(ref_shims.REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == 'source' or k.startswith('source.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_query_range_parts_are_assembled_once_and_in_order(dropin_source, tmp_path):
    """P2S_SHARD=queries: each rank leaves its ordered piece; once the set is complete exactly one rank wins the
    atomic claim and writes the reference's output files, pieces concatenated in rank order, part files removed --
    and a second run into the same directory is not blocked by leftovers (no lock files)"""
    import numpy as np
    ev, _ = dropin_source
    out = str(tmp_path / 'rec')
    world = 3
    rng = np.random.default_rng(0)
    for run in range(2):
        pieces = [(rng.standard_normal(n).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)) for n in (5, 0, 7)]
        for rank in (2, 0):                                   # out of order, one rank still missing
            ev._write_part(out, 'shapeA', rank, *pieces[rank])
            assert ev._try_assemble(out, 'shapeA', world, rank) is False
        ev._write_part(out, 'shapeA', 1, *pieces[1])
        assert ev._try_assemble(out, 'shapeA', world, 1) is True
        assert ev._try_assemble(out, 'shapeA', world, 0) is False          # parts gone
        sdf = np.load(os.path.join(out, 'dist_ms', 'shapeA.xyz.npy'))
        q = np.load(os.path.join(out, 'query_pts_ms', 'shapeA.xyz.npy'))
        assert np.array_equal(sdf, np.concatenate([p[0] for p in pieces]))
        assert np.array_equal(q, np.concatenate([p[1] for p in pieces]))
        assert np.array_equal(np.load(os.path.join(out, 'eval', 'shapeA.xyz.npy')), sdf)
        assert os.listdir(os.path.join(out, '.parts')) == []
        # the visualisations of the file contract (sdf.visualize_query_points) are written without trimesh
        from points2surf_amd import ply
        v, f = ply.read_ply(os.path.join(out, 'query_pts_ms_vis', 'shapeA.ply'))
        assert np.array_equal(v.astype(np.float32), q) and f.shape[0] == 0
        assert os.path.isfile(os.path.join(out, 'vis', 'shapeA.ply'))
