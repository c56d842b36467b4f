"""TEST / BASELINE INFRASTRUCTURE ONLY -- torch-CPU port of the reference's hot path.

Same algorithm as oracle/p2s_oracle.py, written with the ATen CPU ops the reference itself calls
(F.conv1d / F.batch_norm / F.linear / torch.bmm / max over points) and its third-party data-path
calls (scipy cKDTree.query, numpy legacy RandomState), so that timing it on the GPU box's host
cores stands in for "the reference's CPU path (full_eval.py, torch.device('cpu'))", which cannot
travel to the GPU box.  Used by bench.py's ``cpu_baseline`` leg (kind = "port") and checked against
the reference's golden vectors in tests/test_oracle_golden.py.  Never imported by the product.

Follows reference source/points_to_surf_model.py:41-69,100-131,177-234,296-352;
source/data_loader.py:322-421; source/base/utils.py:13-46,62-88,196-227; source/sdf_nn.py:11-21.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _t(w, name):
    return torch.from_numpy(np.ascontiguousarray(w[name]))


class TorchPort:
    def __init__(self, w, cfg):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
        self.cfg = dict(cfg)

    # -- layers --------------------------------------------------------------------------------
    def _conv_bn(self, x, conv, bn, relu=True):
        w = self.w
        x = F.conv1d(x, w[conv + '.weight'], w[conv + '.bias'])
        x = F.batch_norm(x, w[bn + '.running_mean'], w[bn + '.running_var'], w[bn + '.weight'], w[bn + '.bias'],
                         False, 0.1, 1e-5)
        return F.relu(x) if relu else x

    def _fc_bn(self, x, fc, bn):
        w = self.w
        x = F.linear(x, w[fc + '.weight'], w[fc + '.bias'])
        x = F.batch_norm(x, w[bn + '.running_mean'], w[bn + '.running_var'], w[bn + '.weight'], w[bn + '.bias'],
                         False, 0.1, 1e-5)
        return F.relu(x)

    def _trunk(self, x, pre):
        x = self._conv_bn(x, pre + '.conv1', pre + '.bn1')
        x = self._conv_bn(x, pre + '.conv2', pre + '.bn2')
        x = self._conv_bn(x, pre + '.conv3', pre + '.bn3')
        x = x.max(dim=2)[0]
        x = self._fc_bn(x, pre + '.fc1', pre + '.bn4')
        x = self._fc_bn(x, pre + '.fc2', pre + '.bn5')
        return F.linear(x, self.w[pre + '.fc3.weight'], self.w[pre + '.fc3.bias'])

    def _qstn(self, x, pre):
        q = self._trunk(x, pre) + x.new_tensor([1, 0, 0, 0])
        s = 2 / torch.sum(q.pow(2), 1)
        h = torch.bmm(q.unsqueeze(2), q.unsqueeze(1))
        out = q.new_empty(q.shape[0], 3, 3)
        out[:, 0, 0] = 1 - (h[:, 2, 2] + h[:, 3, 3]).mul(s)
        out[:, 0, 1] = (h[:, 1, 2] - h[:, 3, 0]).mul(s)
        out[:, 0, 2] = (h[:, 1, 3] + h[:, 2, 0]).mul(s)
        out[:, 1, 0] = (h[:, 1, 2] + h[:, 3, 0]).mul(s)
        out[:, 1, 1] = 1 - (h[:, 1, 1] + h[:, 3, 3]).mul(s)
        out[:, 1, 2] = (h[:, 2, 3] - h[:, 1, 0]).mul(s)
        out[:, 2, 0] = (h[:, 1, 3] - h[:, 2, 0]).mul(s)
        out[:, 2, 1] = (h[:, 2, 3] + h[:, 1, 0]).mul(s)
        out[:, 2, 2] = 1 - (h[:, 1, 1] + h[:, 2, 2]).mul(s)
        return out

    def _feat(self, x, pre):
        x = self._conv_bn(x, pre + '.conv0a', pre + '.bn0a')
        x = self._conv_bn(x, pre + '.conv0b', pre + '.bn0b')
        t = self._trunk(x, pre + '.stn2')
        t = (t + torch.eye(64).view(1, 4096)).view(-1, 64, 64)
        x = torch.bmm(t, x)
        x = self._conv_bn(x, pre + '.conv1', pre + '.bn1')
        x = self._conv_bn(x, pre + '.conv2', pre + '.bn2')
        x = self._conv_bn(x, pre + '.conv3', pre + '.bn3', relu=False)
        if self.cfg.get('sym_op', 'max') == 'sum':               # reference source/points_to_surf_model.py:213-214
            return torch.sum(x, 2)
        return x.max(dim=2)[0]

    @torch.no_grad()
    def forward(self, patch_ps, sub_ms, query):
        patch = torch.as_tensor(patch_ps).transpose(1, 2)
        shape = (torch.as_tensor(sub_ms) - torch.as_tensor(query).unsqueeze(1)).transpose(1, 2)
        if self.cfg.get('single_transformer'):
            # one PointNetfeat over cat(patch, sub-sample) (reference source/points_to_surf_model.py:320-323)
            x = torch.cat((patch, shape), dim=2)
            if self.cfg.get('use_point_stn'):
                x = torch.bmm(self._qstn(x, 'feat_local_global.stn1'), x)
            f = self._fc_bn(self._feat(x.contiguous(), 'feat_local_global'), 'fc1_local_global', 'bn1_local_global')
            f = self._fc_bn(f, 'fc2', 'bn2')
            f = self._fc_bn(f, 'fc3', 'bn3')
            return F.linear(f, self.w['fc4.weight'], self.w['fc4.bias'])
        if self.cfg.get('use_point_stn') and self.cfg.get('shared_transformer'):
            r = self._qstn(torch.cat((patch, shape), dim=2), 'point_stn')
            shape = torch.bmm(r, shape)
            patch = torch.bmm(r, patch)
        elif self.cfg.get('use_point_stn'):
            # the QSTN of feat_global sees the sub-sample only; the patch is turned like it
            # (reference source/points_to_surf_model.py:177-185, :337-339)
            r = self._qstn(shape, 'feat_global.stn1')
            shape = torch.bmm(r, shape)
            patch = torch.bmm(r, patch)
        g = self._fc_bn(self._feat(shape.contiguous(), 'feat_global'), 'fc1_global', 'bn1_global')
        l = self._fc_bn(self._feat(patch.contiguous(), 'feat_local'), 'fc1_local', 'bn1_local')
        f = torch.cat((l, g), dim=1)
        f = self._fc_bn(f, 'fc2', 'bn2')
        f = self._fc_bn(f, 'fc3', 'bn3')
        return F.linear(f, self.w['fc4.weight'], self.w['fc4.bias'])

    # -- whole path for a list of queries (a3..a9), reference-style per-query data path -------------
    def infer_queries(self, cloud, queries, rng, batch=500):
        """rng: np.random.RandomState (the dataset-wide sub-sample stream). Returns sdf [Q] float32."""
        import scipy.spatial as spatial
        from . import p2s_oracle as O
        cloud = np.ascontiguousarray(cloud, dtype=np.float32)
        tree = spatial.cKDTree(cloud, 1000)                                  # data_loader.py:40-42
        k, n = int(self.cfg.get('points_per_patch', 300)), int(self.cfg.get('sub_sample_size', 1000))
        out = []
        for s in range(0, queries.shape[0], batch):
            qb = np.ascontiguousarray(queries[s:s + batch], dtype=np.float32)
            patch = np.empty((qb.shape[0], k, 3), np.float32)
            rad = np.empty((qb.shape[0],), np.float32)
            sub = np.empty((qb.shape[0], n, 3), np.float32)
            for i in range(qb.shape[0]):                                     # __getitem__, one query at a time
                _, ids = tree.query(x=qb[i], k=k)
                r, ps = O.patch_radius_and_ps(cloud, np.asarray(ids, dtype=np.int32), qb[i])
                patch[i], rad[i] = ps, r
                if self.cfg.get('uniform_subsample'):
                    sid = rng.randint(low=0, high=cloud.shape[0], size=n)
                else:
                    sid = rng.choice(cloud.shape[0], size=n, replace=False, p=O.dist_prob(cloud, qb[i]))
                sub[i] = cloud[sid]
            logits = self.forward(patch, sub, qb).numpy()
            out.append(O.post_process(logits, rad))
        return np.concatenate(out) if out else np.zeros((0,), np.float32)
