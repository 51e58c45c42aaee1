"""Pins the CPU oracle (oracle/cistgcn_ref.py) to vectors produced by the real reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import CASES, assert_close, grad_summary, load_case, make_cfg, state_of
from oracle import cistgcn_ref as O

# parameter gradients live on very different scales (BN-fed conv biases are analytically 0):
# compare each against the global gradient scale of the model.
def _check_attrs(net, rec, mode):
    for k, ref in rec.items():
        if not k.startswith(mode + "/attr/"):
            continue
        obj = net
        for part in k[len(mode + "/attr/"):].split("."):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        got = obj.detach()
        assert_close(got[: ref.shape[0]] if got.shape[0] != ref.shape[0] else got, ref, k)


def _build(name, rec):
    C, T, V, B = [int(v) for v in rec["meta"]]
    net = O.CISTGCN(*make_cfg(C, T, V))
    net.load_state_dict(state_of(rec), strict=True)
    return net


@pytest.mark.parametrize("name", CASES)
def test_state_dict_keys_and_shapes(name):
    rec = load_case(name)
    net = _build(name, rec)
    sd = net.state_dict()
    ref = state_of(rec)
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == tuple(ref[k].shape) for k in sd)


@pytest.mark.parametrize("name", CASES)
def test_eval_forward_and_input_grad(name):
    rec = load_case(name)
    net = _build(name, rec).eval()
    x = torch.from_numpy(rec["x"]).requires_grad_(True)
    pred, = net(x)
    loss = O.mpjpe(pred, torch.from_numpy(rec["target"]))
    loss.backward()
    assert_close(pred, rec["eval/pred"], "pred")
    assert_close(loss, rec["eval/loss"], "loss")
    assert_close(x.grad, rec["eval/dx"], "dL/dx", floor=float(np.abs(rec["eval/dx"]).max()))
    _check_attrs(net, rec, "eval")


@pytest.mark.parametrize("name", CASES)
def test_train_forward_backward(name):
    rec = load_case(name)
    net = _build(name, rec).train()
    x = torch.from_numpy(rec["x"]).requires_grad_(True)
    pred, = net(x)
    loss = O.mpjpe(pred, torch.from_numpy(rec["target"]))
    loss.backward()
    assert_close(pred, rec["train/pred"], "pred")
    assert_close(loss, rec["train/loss"], "loss")
    assert_close(x.grad, rec["train/dx"], "dL/dx", floor=float(np.abs(rec["train/dx"]).max()))
    _check_attrs(net, rec, "train")
    grads = dict(net.named_parameters())
    full = {k[len("train/grad/"):]: v for k, v in rec.items() if k.startswith("train/grad/")}
    summ = {k[len("train/gradsum/"):]: v for k, v in rec.items() if k.startswith("train/gradsum/")}
    assert set(full) | set(summ) == set(grads)
    for k, ref in full.items():
        assert_close(grads[k].grad, ref, "grad " + k, rel=1e-3, floor=1e-2)
    for k, ref in summ.items():
        got = grad_summary(grads[k].grad)
        scale = max(1e-2, ref[2])          # L2 norm of the reference gradient tensor; floor for analytically-zero grads
        assert np.abs(got[3:] - ref[3:]).max() <= 1e-3 * scale, k
        assert abs(got[2] - ref[2]) <= 1e-3 * scale, k
    after = {k[len("train/state_after/"):]: v for k, v in rec.items() if k.startswith("train/state_after/")}
    sd = net.state_dict()
    assert after
    for k, ref in after.items():
        assert_close(sd[k], ref, "running stat " + k)


def test_unit_scale_train():
    rec = load_case("h36m_c8_t10_v22")
    net = _build("h36m_c8_t10_v22", rec).train()
    x = torch.from_numpy(rec["unit/x"]).requires_grad_(True)
    pred, = net(x)
    loss = O.mpjpe(pred, torch.from_numpy(rec["unit/target"]))
    loss.backward()
    assert_close(pred, rec["unit/pred"], "pred", rel=1e-5)
    assert_close(x.grad, rec["unit/dx"], "dL/dx", floor=float(np.abs(rec["unit/dx"]).max()))


def test_ctor_does_not_mutate_config():
    arch, learn = make_cfg(8, 10, 22)
    O.CISTGCN(arch, learn)
    assert arch.model_params.input_gcn.model_complexity == [8] * 4
    assert arch.model_params.output_gcn.model_complexity == [3]


def test_eval_postprocess_matches_reference():
    """oracle/eval_ref.py against vectors produced by the reference's own `_predict` + `losses.mpjpe` (tools/gen_golden_eval.py)"""
    from oracle import eval_ref as E
    rec = load_case("eval_h36m")
    t = lambda k: torch.from_numpy(rec[k])
    used, r22, r32 = rec["dim_used"].tolist(), rec["rep22"].tolist(), rec["rep32"].tolist()
    assert torch.equal(E.gather_used(t("inputs"), used), t("model_input"))
    full = E.scatter_prediction(t("model_output"), t("target"), used, r32, r22)
    assert torch.equal(full, t("predicted_full"))
    assert_close(E.mpjpe_frames(full, t("target")), t("mpjpe_frames"), "per-frame MPJPE", rel=1e-6)
