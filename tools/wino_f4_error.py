#!/usr/bin/env python3
"""Bounded experiment: would Winograd F(4x4, 3x3) hold the parity bar?  CPU emulation in fp32 of the three
transforms and the channel contraction (the arithmetic an MFMA kernel would perform: fp32 products, fp32 accumulation) against an
fp64 direct convolution, on the six VGG16 conv shapes (channel counts and map sizes as in config 2; batch 2 -- the error depends on
the contraction length C, not on the batch).  Inputs as in the network: x = relu(N(0,1)) (what a BatchNorm -> ReLU hands the next
conv), W = kaiming-normal fan_out (models/vgg.py:59-63).  Prints max |err| / max |y| ("of the output scale") for the direct fp32
conv, F(2x2, 3x3) (what the kernels run today) and F(4x4, 3x3).  Acceptance bar set by the judge: <= 2e-5.
Default: the contraction over channels is accumulated SEQUENTIALLY in one fp32 accumulator, as an MFMA chain does; --blocked
uses the CPU GEMM's blocked sums instead (several times more accurate, NOT what a kernel would do)."""
import sys

import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(8)

BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1.]])
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1.]])
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])


def winograd(x, w, m, BT, G, AT):
    """fp32 Winograd F(m x m, 3 x 3), stride 1 pad 1; H, W multiples of m."""
    N, C, H, W = x.shape
    K = w.shape[0]
    a = m + 2
    xp = F.pad(x, (1, 1, 1, 1))
    # tiles [N, C, th, tw, a, a]
    t = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum('ij,ncyxjk,lk->ncyxil', BT, t, BT)          # B^T d B
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)                # G g G^T
    if SEQ:
        # the MFMA's association: one fp32 accumulator per output, channels added one after the other (v_mfma_f32_32x32x2_f32 is
        # an fmaf chain, MI355X guide) -- the blocked sums of a CPU GEMM are several times more accurate on long contractions
        M = torch.zeros(N, K, V.shape[2], V.shape[3], a, a)
        for c in range(C):
            M.addcmul_(U[None, :, c, None, None], V[:, None, c])
    else:
        M = torch.einsum('kcij,ncyxij->nkyxij', U, V)            # sum over channels per position (fp32, blocked)
    Y = torch.einsum('ij,nkyxjl,ml->nkyxim', AT, M, AT)          # A^T M A -> [N,K,th,tw,m,m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def direct_seq(x, w):
    """direct conv with the kernels' association: one accumulator, (channel, tap) terms added sequentially"""
    N, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    y = torch.zeros(N, w.shape[0], H, W)
    for c in range(C):
        for r in range(3):
            for q in range(3):
                y.addcmul_(w[None, :, c, r, q, None, None], xp[:, None, c, r:r + H, q:q + W])
    return y


SEQ = '--blocked' not in sys.argv        # --randn: zero-mean inputs (what tests/test_hip_parity.py::test_winograd_matches_direct feeds)


def main():
    shapes = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)]
    print('%-18s %12s %12s %12s' % ('C->K @HxW', 'direct fp32', 'F(2x2,3x3)', 'F(4x4,3x3)'))
    worst = [0, 0, 0]
    for C, K, H in shapes:
        Hs = min(H, 28) // 4 * 4                                      # crop large maps (tiles are independent): keeps the run bounded
        N = 2
        x = torch.randn(N, C, Hs, Hs) if '--randn' in sys.argv else torch.relu(torch.randn(N, C, Hs, Hs))
        w = torch.randn(K, C, 3, 3) * (2.0 / (9 * K)) ** 0.5
        ref = F.conv2d(x.double(), w.double(), padding=1)
        scale = float(ref.abs().max())
        errs = [float(((direct_seq(x, w) if SEQ else F.conv2d(x, w, padding=1)).double() - ref).abs().max()) / scale,
                float((winograd(x, w, 2, BT2, G2, AT2).double() - ref).abs().max()) / scale,
                float((winograd(x, w, 4, BT4, G4, AT4).double() - ref).abs().max()) / scale]
        worst = [max(a, b) for a, b in zip(worst, errs)]
        print('%-18s %12.2e %12.2e %12.2e' % ('%d->%d @%d' % (C, K, H), *errs), flush=True)
    print('%-18s %12.2e %12.2e %12.2e' % ('worst', *worst))
    print('bar 2e-5: F(4x4,3x3) %s' % ('PASSES' if worst[2] <= 2e-5 else 'FAILS'))


if __name__ == '__main__':
    sys.exit(main())
