"""Parameter/buffer layout of the Points2Surf network -- the on-disk contract.

The reference saves ``DataParallel(PointsToSurfModel).state_dict()`` (reference
source/points_to_surf_train.py:513) and re-loads it at inference
(source/points_to_surf_eval.py:150-171).  This module states that key/shape layout
(reference source/points_to_surf_model.py:12-131 STN/QSTN, :134-176 PointNetfeat,
:237-294 PointsToSurfModel) as data, so that the engine can fold / repack weights
and the drop-in ``nn.Module`` can expose identical names.
"""
from collections import OrderedDict

MODEL_DEFAULTS = dict(
    net_size_max=1024, num_points=300, output_dim=2, use_point_stn=False, use_feat_stn=True,
    sym_op='max', use_query_point=True, sub_sample_size=1000, do_augmentation=False,
    single_transformer=False, shared_transformation=False)

# hyper-parameters of the two named models (reference experiments/train_p2s_max.sh:15-33,
# experiments/train_p2s_vanilla.sh:15-32)
NAMED_MODELS = {
    'p2s_max': dict(use_point_stn=False, shared_transformation=False, uniform_subsample=True),
    'p2s_vanilla': dict(use_point_stn=True, shared_transformation=True, uniform_subsample=False),
    # ablation models of the paper (experiments/train_p2s_{uniform,no_qstn,small_kNN,large_kNN}.sh): the QSTN lives in
    # feat_global (sees the sub-sample only), its rotation is applied to the sub-sample and to the patch
    # (source/points_to_surf_model.py:283-284, :337-339)
    'p2s_uniform': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=True),
    'p2s_no_qstn': dict(use_point_stn=False, shared_transformation=False, uniform_subsample=False),
    'p2s_small_kNN': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, points_per_patch=75),
    'p2s_large_kNN': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, points_per_patch=1200),
    # experiments/train_p2s_regression.sh: ONE output, the signed distance itself ('imp_surf': tanh^2 * sign, sdf_nn.py:6-8)
    'p2s_regression': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, output_dim=1),
    # experiments/train_p2s_shared_encoder.sh (--single_transformer 1): one PointNetfeat over cat(patch, sub-sample)
    # with its own QSTN + STN, then fc1_local_global 1024 -> 1024 (source/points_to_surf_model.py:253-263, :320-323)
    'p2s_shared_encoder': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, single_transformer=True),
    # experiments/train_p2s_{small,medium,large}_radius.sh (--patch_radius 0.05 / 0.1 / 0.2): the patch is every point within
    # that distance of the query -- a random 300 of them if there are more, padded with the query point if fewer
    # (source/base/point_cloud.py:177-191)
    # train --use_feat_stn 0 (no script of the reference sets it): p2s_max without the 64x64 feature transforms
    'p2s_max_no_feat_stn': dict(use_point_stn=False, shared_transformation=False, uniform_subsample=True, use_feat_stn=False),
    # train --sym_op sum (set by no script of the reference): PointNetfeat pools with a sum over the points
    # (source/points_to_surf_model.py:170-175, :211-214); the STN / QSTN trunks keep their max-pool
    'p2s_max_sum': dict(use_point_stn=False, shared_transformation=False, uniform_subsample=True, sym_op='sum'),
    'p2s_shared_encoder_sum': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False,
                                   single_transformer=True, sym_op='sum'),
    'p2s_small_radius': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, patch_radius=0.05),
    'p2s_medium_radius': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, patch_radius=0.1),
    'p2s_large_radius': dict(use_point_stn=True, shared_transformation=False, uniform_subsample=False, patch_radius=0.2),
}


def _bn(prefix, c, out):
    out[prefix + '.weight'] = (c,)
    out[prefix + '.bias'] = (c,)
    out[prefix + '.running_mean'] = (c,)
    out[prefix + '.running_var'] = (c,)
    out[prefix + '.num_batches_tracked'] = ()


def _conv(prefix, cin, cout, out):
    out[prefix + '.weight'] = (cout, cin, 1)
    out[prefix + '.bias'] = (cout,)


def _fc(prefix, cin, cout, out):
    out[prefix + '.weight'] = (cout, cin)
    out[prefix + '.bias'] = (cout,)


def _trunk(prefix, dim_in, n, n_out, out):
    """STN (n_out = dim*dim) and QSTN (n_out = 4) share this layout; registration order
    follows the reference constructors (convs, fcs, then bns)."""
    _conv(prefix + '.conv1', dim_in, 64, out)
    _conv(prefix + '.conv2', 64, 128, out)
    _conv(prefix + '.conv3', 128, n, out)
    _fc(prefix + '.fc1', n, n // 2, out)
    _fc(prefix + '.fc2', n // 2, n // 4, out)
    _fc(prefix + '.fc3', n // 4, n_out, out)
    _bn(prefix + '.bn1', 64, out)
    _bn(prefix + '.bn2', 128, out)
    _bn(prefix + '.bn3', n, out)
    _bn(prefix + '.bn4', n // 2, out)
    _bn(prefix + '.bn5', n // 4, out)


def _pointnetfeat(prefix, n, use_point_stn, use_feat_stn, out):
    if use_point_stn:
        _trunk(prefix + '.stn1', 3, n, 4, out)
    if use_feat_stn:
        _trunk(prefix + '.stn2', 64, n, 64 * 64, out)
    _conv(prefix + '.conv0a', 3, 64, out)
    _conv(prefix + '.conv0b', 64, 64, out)
    _bn(prefix + '.bn0a', 64, out)
    _bn(prefix + '.bn0b', 64, out)
    _conv(prefix + '.conv1', 64, 64, out)
    _conv(prefix + '.conv2', 64, 128, out)
    _conv(prefix + '.conv3', 128, n, out)
    _bn(prefix + '.bn1', 64, out)
    _bn(prefix + '.bn2', 128, out)
    _bn(prefix + '.bn3', n, out)


def state_shapes(net_size_max=1024, output_dim=2, use_point_stn=False, use_feat_stn=True,
                 single_transformer=False, shared_transformation=False, **_unused):
    """Ordered {name: shape} of every parameter and buffer (no ``module.`` prefix)."""
    n = int(net_size_max)
    out = OrderedDict()
    if single_transformer:
        _pointnetfeat('feat_local_global', n, bool(use_point_stn), use_feat_stn, out)
        _fc('fc1_local_global', n, n, out)
        _bn('bn1_local_global', n, out)
    else:
        if use_point_stn and shared_transformation:
            _trunk('point_stn', 3, n, 4, out)
        _pointnetfeat('feat_local', n, False, use_feat_stn, out)
        _pointnetfeat('feat_global', n, bool(use_point_stn and not shared_transformation), use_feat_stn, out)
        _fc('fc1_local', n, n // 2, out)
        _fc('fc1_global', n, n // 2, out)
        _bn('bn1_local', n // 2, out)
        _bn('bn1_global', n // 2, out)
    _fc('fc2', n, n // 4, out)
    _fc('fc3', n // 4, n // 8, out)
    _fc('fc4', n // 8, output_dim, out)
    _bn('bn2', n // 4, out)
    _bn('bn3', n // 8, out)
    return out


def strip_module_prefix(state_dict):
    """``module.`` comes from saving the DataParallel wrapper (reference points_to_surf_train.py:513)."""
    return {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
