#!/usr/bin/env python3
"""Generate golden vectors for the CIST-GCN hot path from the REAL reference implementation.

Runs only in the build container, where the upstream sources are mounted read-only at
/root/reference.  Nothing of the reference travels: this script imports it, runs it on seeded
inputs and writes *data* (inputs, weights, outputs, gradients) to tests/golden/*.npz.

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference package's __init__ pulls in plotting/FLOP-count dependencies that are not in the
image (fvcore); the model file itself only needs torch, so the package is pre-registered as an
empty namespace and only `models.CISTGCN.CISTGCN` is imported.
"""
import importlib
import os
import sys
import types
from types import SimpleNamespace as NS

import numpy as np
import torch

REF_ROOT = "/root/reference"
OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# name -> (C, T_in, V, B).  B >= 4: several train-mode BatchNorms normalise over the batch axis only
# (SURVEY appendix 'Minimum batch'); with 2-3 samples their backward is numerically degenerate.
CASES = {
    "h36m_c8_t10_v22": (8, 10, 22, 4),     # reference-YAML H3.6M shape (train_h36m.yaml:4-6)
    "h36m_c8_t50_v22": (8, 50, 22, 4),     # BASELINE.json configs[0]/[1] shape
    "amass_c16_t10_v18": (16, 10, 18, 4),  # reference-YAML AMASS joints (train_amass.yaml:5)
    "cmu_c8_t50_v25": (8, 50, 25, 4),      # BASELINE.json "25-joint" shape
}
FULL_GRADS = {"h36m_c8_t10_v22"}      # these also keep the unit-scale run and the Adj maps of every block


class BranchRecorder:
    """Forward hooks on every nn.PReLU of the reference: the branch each element took (input > 0), bit-packed.  The train-mode
    fixtures sit on PReLU kinks (batch-statistic BatchNorm over 4 samples): another summation order - a different OMP thread count is
    enough - lands a rounding-sized pre-activation on the other side of 0 and moves whole gradient tensors by percents.  With the
    branches recorded, a checker differentiates the SAME piecewise-linear function and the fp32 bound applies to every gradient."""

    def __init__(self, net):
        self.bits, self.handles = {}, []
        for name, m in net.named_modules():
            if isinstance(m, torch.nn.PReLU):
                self.handles.append(m.register_forward_hook(lambda mod, inp, out, name=name: self.bits.__setitem__(name, inp[0].detach() > 0)))

    def close(self):
        for h in self.handles:
            h.remove()
        return {k: np.packbits(v.numpy().reshape(-1)) for k, v in self.bits.items()}


def import_reference():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("human_motion_prediction")
    pkg.__path__ = [os.path.join(REF_ROOT, "human_motion_prediction")]
    sys.modules["human_motion_prediction"] = pkg
    model = importlib.import_module("human_motion_prediction.models.CISTGCN.CISTGCN")
    losses = importlib.import_module("human_motion_prediction.losses.losses")
    return model, losses


def make_cfg(C, T, V, dropout=0.0):
    arch = NS(model_params=NS(
        input_n=T, output_n=25, joints=V, n_txcnn_layers=4, txc_kernel_size=3, reduction=8,
        hidden_dim=64, clipping=15,
        input_gcn=NS(model_complexity=[C] * 4, interpretable=[True] * 5),
        output_gcn=NS(model_complexity=[3], interpretable=[True])))
    return arch, NS(dropout=dropout)


def randomise(net, gen):
    """Move every tensor away from its init so that no path is numerically silent
    (fresh Map2Adj weights give Adj ~ 1e-8 in eval mode)."""
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.add_(0.5 * torch.randn(p.shape, generator=gen) / fan_in ** 0.5)
            elif name.endswith("bias"):
                p.add_(0.2 * torch.randn(p.shape, generator=gen))
            else:   # BN gamma, PReLU alpha
                p.add_(0.1 * torch.randn(p.shape, generator=gen))


def calibrate_running_stats(net, x, gen):
    """One train-mode pass with momentum 1 so that running stats describe mm-scale data, then a
    mild perturbation so eval-mode BN differs from batch-stat BN."""
    bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    for m in bns:
        m.momentum = 1.0
    net.train()
    with torch.no_grad():
        net(x)
        for m in bns:
            m.momentum = 0.1
            m.running_var.mul_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=gen))
            m.running_mean.add_(0.05 * m.running_var.sqrt() * torch.randn(m.running_mean.shape, generator=gen))
            m.num_batches_tracked.zero_()


def attrs(net, nblocks):
    out = {}
    blocks = [("st_gcnns.%d" % i, net.st_gcnns[i]) for i in nblocks] + [("st_gcnns_o.0", net.st_gcnns_o[0])]
    for name, blk in blocks:
        out[name + ".dsgn.Adj"] = blk.dsgn.Adj[:1]
        out[name + ".tsgn.Adj"] = blk.tsgn.Adj[:1]
    for i, blk in enumerate(net.st_gcnns):
        out["st_gcnns.%d.w1" % i], out["st_gcnns.%d.w2" % i] = blk.w1, blk.w2
    out["st_gcnns_o.0.w1"], out["st_gcnns_o.0.w2"] = net.st_gcnns_o[0].w1, net.st_gcnns_o[0].w2
    c = net.context_layer
    for k in ("joints", "displacements", "seq_joints", "seq_joints_n", "seq_joints_dims"):
        out["context_layer." + k] = getattr(c, k)
    return {k: v.detach().numpy().copy() for k, v in out.items()}


def grad_summary(g):
    f = g.flatten().double()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.norm().item()], f[:61].numpy()]).astype(np.float64)


def main():
    ref_model, ref_losses = import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    for name, (C, T, V, B) in CASES.items():
        gen = torch.Generator().manual_seed(sum(map(ord, name)))
        torch.manual_seed(0)
        arch, learn = make_cfg(C, T, V)
        net = ref_model.CISTGCN(arch, learn)
        randomise(net, gen)
        calibrate_running_stats(net, 50 + 350 * torch.randn(8, T, V, 3, generator=gen), gen)
        rec = {"meta": np.array([C, T, V, B], dtype=np.int64)}
        for k, v in net.state_dict().items():
            rec["state/" + k] = v.numpy().copy()

        x = 50 + 350 * torch.randn(B, T, V, 3, generator=gen)
        tgt = x[:, -1:] + 20 * torch.randn(B, 25, V, 3, generator=gen)
        rec["x"], rec["target"] = x.numpy(), tgt.numpy()
        blocks_with_adj = range(5) if name in FULL_GRADS else (1,)

        # ---- eval mode (running-stat BN): forward, attrs, dL/dx (adversarial path) ----
        net.eval()
        net.zero_grad()
        xe = x.clone().requires_grad_(True)
        pred, = net(xe)
        loss = ref_losses.mpjpe(pred, tgt)
        loss.backward()
        rec["eval/pred"], rec["eval/loss"], rec["eval/dx"] = pred.detach().numpy(), loss.detach().numpy(), xe.grad.numpy().copy()
        for k, v in attrs(net, blocks_with_adj).items():
            rec["eval/attr/" + k] = v

        # ---- train mode, dropout 0 (batch-stat BN): forward, loss, all grads, running-stat update ----
        net.train()
        net.zero_grad()
        xt = x.clone().requires_grad_(True)
        recorder = BranchRecorder(net)
        pred, = net(xt)
        for k, v in recorder.close().items():
            rec["train/branch/" + k] = v
        loss = ref_losses.mpjpe(pred, tgt)
        loss.backward()
        rec["train/pred"], rec["train/loss"], rec["train/dx"] = pred.detach().numpy(), loss.detach().numpy(), xt.grad.numpy().copy()
        for k, v in attrs(net, blocks_with_adj).items():
            rec["train/attr/" + k] = v
        for k, p in net.named_parameters():
            rec["train/grad/" + k] = p.grad.numpy().copy()            # every gradient of every case (round 4; was: 64-element summaries)
            if name not in FULL_GRADS:
                rec["train/gradsum/" + k] = grad_summary(p.grad)
        for k, v in net.state_dict().items():
            if "running_" in k and (k.startswith("st_gcnns.1.") or k.startswith("st_gcnns_o.0.ts") or k.startswith("context_layer.f")):
                rec["train/state_after/" + k] = v.numpy().copy()

        # ---- unit-scale input, train mode (tolerance evidence at small magnitudes) ----
        if name in FULL_GRADS:
            xu = torch.randn(B, T, V, 3, generator=gen)
            tu = xu[:, -1:] + 0.1 * torch.randn(B, 25, V, 3, generator=gen)
            net.zero_grad()
            xu_ = xu.clone().requires_grad_(True)
            pred, = net(xu_)
            loss = ref_losses.mpjpe(pred, tu)
            loss.backward()
            rec["unit/x"], rec["unit/target"] = xu.numpy(), tu.numpy()
            rec["unit/pred"], rec["unit/loss"], rec["unit/dx"] = pred.detach().numpy(), loss.detach().numpy(), xu_.grad.numpy().copy()

        # ---- the same train-mode step by the reference in fp64 (round 4): the ground truth of this fixture.  An fp64 restatement of the
        # algorithm on the fp64 reference's branches must reproduce it to ~1e-10 whatever the thread count (the algorithmic pin proper);
        # |fp32 reference - fp64 reference| per tensor is the noise floor of every fp32 comparison against this fixture.
        arch64, learn64 = make_cfg(C, T, V)
        net64 = ref_model.CISTGCN(arch64, learn64).double()      # (built last: its init draws from the global RNG, nothing above may move)
        net64.load_state_dict({k: torch.from_numpy(v).double() if v.dtype.kind == "f" else torch.from_numpy(v) for k, v in
                               ((k[len("state/"):], rec[k]) for k in rec if k.startswith("state/"))})
        net64.train()
        net64.zero_grad()
        x64 = x.double().requires_grad_(True)
        recorder = BranchRecorder(net64)
        pred64, = net64(x64)
        for k, v in recorder.close().items():
            rec["train64/branch/" + k] = v
        loss64 = ref_losses.mpjpe(pred64, tgt.double())
        loss64.backward()
        rec["train64/pred"], rec["train64/loss"], rec["train64/dx"] = pred64.detach().numpy(), loss64.detach().numpy(), x64.grad.numpy().copy()
        for k, p in net64.named_parameters():
            rec["train64/grad/" + k] = p.grad.numpy().copy()
        for k, v in attrs(net64, blocks_with_adj).items():
            rec["train64/attr/" + k] = v

        path = os.path.join(OUT_DIR, name + ".npz")
        np.savez_compressed(path, **rec)
        print("%-22s %4d arrays  %.2f MB  |pred|max %.1f  loss(train) %.3f" % (
            name, len(rec), os.path.getsize(path) / 1e6, float(np.abs(rec["train/pred"]).max()), float(rec["train/loss"])))


if __name__ == "__main__":
    main()
