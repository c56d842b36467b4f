"""Drop-in ``source.points_to_surf_model`` (boundary level B2).

Same constructor keywords, same parameter / buffer names (so that
``DataParallel(m).load_state_dict(torch.load(f))`` of a reference checkpoint succeeds, reference
source/points_to_surf_eval.py:167-170) and same ``forward(dict) -> [B, output_dim]`` contract as the
reference ``PointsToSurfModel`` (source/points_to_surf_model.py:237-352) -- but ``forward`` runs on
the HIP engine (libp2s_hip.so) instead of ~95 eager ATen kernels.  The module holds the parameters
only as the checkpoint container; there is no PyTorch compute path and no CPU fallback.
"""
import torch
import torch.nn as nn

from points2surf_amd import engine as _engine

input_dims_per_point = 3


def _trunk(dim_in, n, n_out):
    m = nn.Module()
    m.conv1 = nn.Conv1d(dim_in, 64, 1)
    m.conv2 = nn.Conv1d(64, 128, 1)
    m.conv3 = nn.Conv1d(128, n, 1)
    m.fc1 = nn.Linear(n, n // 2)
    m.fc2 = nn.Linear(n // 2, n // 4)
    m.fc3 = nn.Linear(n // 4, n_out)
    m.bn1 = nn.BatchNorm1d(64)
    m.bn2 = nn.BatchNorm1d(128)
    m.bn3 = nn.BatchNorm1d(n)
    m.bn4 = nn.BatchNorm1d(n // 2)
    m.bn5 = nn.BatchNorm1d(n // 4)
    return m


def _encoder(n, use_point_stn, use_feat_stn):
    m = nn.Module()
    if use_point_stn:
        m.stn1 = _trunk(3, n, 4)
    if use_feat_stn:
        m.stn2 = _trunk(64, n, 64 * 64)
    m.conv0a = nn.Conv1d(input_dims_per_point, 64, 1)
    m.conv0b = nn.Conv1d(64, 64, 1)
    m.bn0a = nn.BatchNorm1d(64)
    m.bn0b = nn.BatchNorm1d(64)
    m.conv1 = nn.Conv1d(64, 64, 1)
    m.conv2 = nn.Conv1d(64, 128, 1)
    m.conv3 = nn.Conv1d(128, n, 1)
    m.bn1 = nn.BatchNorm1d(64)
    m.bn2 = nn.BatchNorm1d(128)
    m.bn3 = nn.BatchNorm1d(n)
    return m


class PointsToSurfModel(nn.Module):
    def __init__(self, net_size_max=1024, num_points=500, output_dim=3, use_point_stn=True, use_feat_stn=True,
                 sym_op='max', use_query_point=False,
                 sub_sample_size=500, do_augmentation=True, single_transformer=False, shared_transformation=False):
        super(PointsToSurfModel, self).__init__()
        if sym_op not in ('max', 'sum'):
            raise ValueError('Unsupported symmetric operation: %s' % sym_op)        # reference :175
        self.sym_op = sym_op
        self.net_size_max = net_size_max
        self.num_points = num_points
        self.output_dim = output_dim
        self.use_query_point = use_query_point
        self.use_point_stn = bool(use_point_stn)
        self.use_feat_stn = bool(use_feat_stn)
        self.sub_sample_size = sub_sample_size
        self.num_query_points = int(self.use_query_point)
        self.do_augmentation = do_augmentation
        self.single_transformer = bool(single_transformer)
        self.shared_transformation = bool(shared_transformation)
        n = int(net_size_max)

        if self.single_transformer:          # reference :253-263: one encoder over cat(patch, sub-sample)
            self.feat_local_global = _encoder(n, self.use_point_stn, use_feat_stn)
            self.fc1_local_global = nn.Linear(n, n)
            self.bn1_local_global = nn.BatchNorm1d(n)
        else:
            if self.use_point_stn and self.shared_transformation:
                self.point_stn = _trunk(3, n, 4)
            self.feat_local = _encoder(n, False, use_feat_stn)
            self.feat_global = _encoder(n, self.use_point_stn and not self.shared_transformation, use_feat_stn)
            self.fc1_local = nn.Linear(n, n // 2)
            self.fc1_global = nn.Linear(n, n // 2)
            self.bn1_local = nn.BatchNorm1d(n // 2)
            self.bn1_global = nn.BatchNorm1d(n // 2)
        self.fc2 = nn.Linear(n, n // 4)
        self.fc3 = nn.Linear(n // 4, n // 8)
        self.fc4 = nn.Linear(n // 8, output_dim)
        self.bn2 = nn.BatchNorm1d(n // 4)
        self.bn3 = nn.BatchNorm1d(n // 8)

        self._engine_model = None
        self._engine_key = None

    # -- engine binding ---------------------------------------------------------------------------
    def engine_cfg(self):
        return dict(net_size=self.net_size_max, points_per_patch=self.num_points,
                    sub_sample_size=self.sub_sample_size, output_dim=self.output_dim,
                    use_point_stn=self.use_point_stn, shared_transformer=self.shared_transformation,
                    use_feat_stn=self.use_feat_stn, single_transformer=self.single_transformer, sym_op=self.sym_op)

    def _engine(self, device):
        # the packed weight blob is rebuilt when the parameters change (load_state_dict, .to(), in-place edits)
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.state_dict().values())
        if self._engine_model is None or key != self._engine_key:
            if self.training:
                raise RuntimeError('the HIP engine implements inference (BatchNorm running statistics): call .eval()')
            self._engine_model = _engine.Model(self.state_dict(), self.engine_cfg(), device=device)
            self._engine_key = key
        return self._engine_model

    def forward(self, x):
        patch = x['patch_pts_ps']
        sub = x['pts_sub_sample_ms']
        query = x['imp_surf_query_point_ms']
        if not patch.is_cuda:
            raise RuntimeError('PointsToSurfModel (MI355X engine) needs CUDA/ROCm tensors; there is no CPU path')
        with torch.no_grad():
            logits, _ = self._engine(patch.device).forward(patch, sub, query, want_logits=True, want_sdf=False)
            # documented side effect of the reference forward (:303): the caller's sub-sample tensor is
            # translated to the query point in place
            sub -= query.unsqueeze(1)
        return logits
