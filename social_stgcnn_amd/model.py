"""Drop-in `model.py`: ConvTemporalGraphical / st_gcn / social_stgcnn on MI355X HIP kernels.

Same class names, constructor signatures, sub-module tree and 40 state_dict keys as the
reference (model.py:18-198), so `from model import *` in train.py / test.py and the shipped
`val_best.pth` files keep working.  forward(v, a) -> (v_out, a).

Differences that are extensions, not changes:
  * `a` may be (T,V,V) like the reference or batched (N,T,V,V) (north-star 'nctv,ntvw->nctw');
  * `num_peds` (int32 (N,)) marks the valid pedestrians of each padded scene;
  * in train mode BatchNorm statistics are PER SCENE (the reference trains with N = 1 per
    forward, train.py:173-177, so a batch of N scenes equals N reference forwards, including the
    N sequential running-stat updates).  For N == 1 this is exactly nn.BatchNorm2d.

Compute never falls back to eager PyTorch: unsupported configurations raise NotImplementedError.
"""
import torch
import torch.nn as nn

from . import ops


class ConvTemporalGraphical(nn.Module):
    r"""Graph convolution: (t_kernel x 1) conv then `einsum('nctv,tvw->nctw')` (model.py:18-68).

    Shape: x (N, in_channels, T, V); A (K, V, V) with K == kernel_size, or (N, K, V, V).
    Returns (contiguous (N, out_channels, T_out, V), A).
    """

    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1, t_padding=0,
                 t_dilation=1, bias=True):
        super(ConvTemporalGraphical, self).__init__()
        self.kernel_size = kernel_size
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=(t_kernel_size, 1),
                              padding=(t_padding, 0), stride=(t_stride, 1), dilation=(t_dilation, 1), bias=bias)

    def forward(self, x, A, num_peds=None):
        assert A.size(-3) == self.kernel_size
        if self.conv.stride != (1, 1) or self.conv.dilation != (1, 1):
            raise NotImplementedError("ConvTemporalGraphical HIP path supports t_stride=1, t_dilation=1")
        x = ops.conv_t(x, self.conv.weight, self.conv.bias, self.conv.padding[0], num_peds)
        x = ops.spatial_agg(x, A, num_peds)
        return x.contiguous(), A


class _Identity:
    def __call__(self, x):
        return x


class _Zero:
    def __call__(self, x):
        return 0


def _block_tensors(blk):
    """(parameters, running-stat buffers, num_batches_tracked) of one st_gcn in kernel order."""
    params = [blk.gcn.conv.weight, blk.gcn.conv.bias, blk.tcn[0].weight, blk.tcn[0].bias, blk.tcn[1].weight,
              blk.tcn[2].weight, blk.tcn[2].bias, blk.tcn[3].weight, blk.tcn[3].bias]
    bufs = [blk.tcn[0].running_mean, blk.tcn[0].running_var, blk.tcn[3].running_mean, blk.tcn[3].running_var]
    nbt = [blk.tcn[0].num_batches_tracked, blk.tcn[3].num_batches_tracked]
    if isinstance(blk.residual, nn.Sequential):
        params += [blk.residual[0].weight, blk.residual[0].bias, blk.residual[1].weight, blk.residual[1].bias]
        bufs += [blk.residual[1].running_mean, blk.residual[1].running_var]
        nbt += [blk.residual[1].num_batches_tracked]
    params += [blk.prelu.weight]
    return params, bufs, nbt


def _check_block(blk, training):
    if blk.gcn.conv.bias is None:
        raise NotImplementedError("st_gcn HIP path needs the gcn conv bias")
    if blk.stride != 1:
        raise NotImplementedError("st_gcn HIP path supports stride=1")
    if training and blk.tcn[4].p > 0:
        raise NotImplementedError("st_gcn HIP path supports dropout=0 in training")
    bn = blk.tcn[0]
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise NotImplementedError("st_gcn HIP path needs default BatchNorm2d settings")


class st_gcn(nn.Module):
    r"""Spatial temporal graph convolution block (model.py:71-155):
    PReLU( BN(Conv_{kt x 1}(PReLU(BN(gcn(x, A))))) + residual(x) ).

    kernel_size = (temporal kernel, graph kernel K = seq_len).  One fused HIP kernel per direction.
    """

    def __init__(self, in_channels, out_channels, kernel_size, use_mdn=False, stride=1, dropout=0, residual=True):
        super(st_gcn, self).__init__()
        assert len(kernel_size) == 2
        assert kernel_size[0] % 2 == 1
        padding = ((kernel_size[0] - 1) // 2, 0)
        self.use_mdn = use_mdn
        self.stride = stride
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kt = kernel_size[0]
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1])
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.PReLU(),
            nn.Conv2d(out_channels, out_channels, (kernel_size[0], 1), (stride, 1), padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        if not residual:
            self.residual = _Zero()
            self.residual_kind = 0
        elif (in_channels == out_channels) and (stride == 1):
            self.residual = _Identity()
            self.residual_kind = 1
        else:
            self.residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels),
            )
            self.residual_kind = 2
        self.prelu = nn.PReLU()
        self._pp = ops.FlatPack()
        self._pb = ops.FlatPack()

    def forward(self, x, A, num_peds=None):
        assert A.size(-3) == self.gcn.kernel_size
        _check_block(self, self.training)
        params, bufs, nbt = _block_tensors(self)
        flat_p = self._pp.ensure(params)
        flat_b = self._pb.ensure(bufs)
        desc = ops.make_desc(1, 0, self.in_channels, self.out_channels, x.shape[2], 0, self.kt, self.residual_kind,
                             self.use_mdn, self.training, self.tcn[0].eps, self.tcn[0].momentum,
                             options=getattr(self, "options", None))
        y = ops.fused_model(x, A, num_peds, desc, flat_p, flat_b, nbt, frozenset(), params, self)
        return y, A


class social_stgcnn(nn.Module):
    """Social-STGCNN (model.py:157-198): n_stgcnn st_gcn blocks, a memory reinterpretation to
    (N, seq_len, C, V), the TXP-CNN 3x3 convolutions with PReLU/residuals, and a reinterpretation to
    (N, C, pred_seq_len, V) -- ONE scene-resident HIP kernel forward, one backward."""

    def __init__(self, n_stgcnn=1, n_txpcnn=1, input_feat=2, output_feat=5, seq_len=8, pred_seq_len=12,
                 kernel_size=3):
        super(social_stgcnn, self).__init__()
        self.n_stgcnn = n_stgcnn
        self.n_txpcnn = n_txpcnn
        self.input_feat = input_feat
        self.output_feat = output_feat
        self.seq_len = seq_len
        self.pred_seq_len = pred_seq_len
        self.kt = kernel_size

        self.st_gcns = nn.ModuleList()
        self.st_gcns.append(st_gcn(input_feat, output_feat, (kernel_size, seq_len)))
        for j in range(1, self.n_stgcnn):
            self.st_gcns.append(st_gcn(output_feat, output_feat, (kernel_size, seq_len)))

        self.tpcnns = nn.ModuleList()
        self.tpcnns.append(nn.Conv2d(seq_len, pred_seq_len, 3, padding=1))
        for j in range(1, self.n_txpcnn):
            self.tpcnns.append(nn.Conv2d(pred_seq_len, pred_seq_len, 3, padding=1))
        self.tpcnn_ouput = nn.Conv2d(pred_seq_len, pred_seq_len, 3, padding=1)

        self.prelus = nn.ModuleList()
        for j in range(self.n_txpcnn):
            self.prelus.append(nn.PReLU())
        self._pp = ops.FlatPack()
        self._pb = ops.FlatPack()

    # ---- packed views -----------------------------------------------------------------------
    def _tensors(self):
        params, bufs, nbt = [], [], []
        for blk in self.st_gcns:
            p, b, c = _block_tensors(blk)
            params += p
            bufs += b
            nbt += c
        n_blk = len(params)
        for conv in self.tpcnns:
            params += [conv.weight, conv.bias]
        params += [self.tpcnn_ouput.weight, self.tpcnn_ouput.bias]
        params += [p.weight for p in self.prelus]
        # parameters forward() never touches (model.py:191 `range(1, n_txpcnn-1)`): grad stays None
        hidden = max(1, self.n_txpcnn - 1)
        dead = set()
        for j in range(hidden, self.n_txpcnn):
            dead.add(n_blk + 2 * j)
            dead.add(n_blk + 2 * j + 1)
            dead.add(n_blk + 2 * self.n_txpcnn + 2 + j)
        return params, bufs, nbt, frozenset(dead)

    def flat_parameters(self):
        """The flat fp32 buffer all parameters are views of (named_parameters() order)."""
        params = self._tensors()[0]
        return self._pp.ensure(params)

    def forward(self, v, a, num_peds=None):
        assert a.size(-3) == self.st_gcns[0].gcn.kernel_size
        if self.n_txpcnn < 1:
            raise NotImplementedError("social_stgcnn needs n_txpcnn >= 1 (model.py:168 always builds tpcnns[0])")
        for blk in self.st_gcns:
            _check_block(blk, self.training)
        params, bufs, nbt, dead = self._tensors()
        flat_p = self._pp.ensure(params)
        flat_b = self._pb.ensure(bufs)
        bn = self.st_gcns[0].tcn[0]
        desc = ops.make_desc(self.n_stgcnn, self.n_txpcnn, self.input_feat, self.output_feat, self.seq_len,
                             self.pred_seq_len, self.kt, self.st_gcns[0].residual_kind, False, self.training,
                             bn.eps, bn.momentum, options=getattr(self, "options", None))
        y = ops.fused_model(v, a, num_peds, desc, flat_p, flat_b, nbt, dead, params, self)
        return y, a
