"""Optimizer set-up (second slice) against fixtures produced by the reference's own `build_custom_optimizer`
(`Detic/detic/custom_solver.py:19-79`; tests/golden/gen_golden_solver.py): parameter groups on the CPU, the update rule of the
recurrent configuration (clip by value + AdamW) through `eod_adamw_step` on the GPU."""
import json
import math
import os

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "solver.json")) as fh:
        return json.load(fh)


class _P:
    def __init__(self, requires_grad=True):
        self.requires_grad = requires_grad


def _named(golden):
    ps = [_P(n not in golden["frozen"]) for n in golden["names"]]
    return list(zip(golden["names"], ps)) + [("alias.of.conv1.weight", ps[0])]


@pytest.mark.parametrize("case", ["recurrent_yaml", "sgd_backbone_multiplier", "adamw_full_model_clip"])
def test_param_groups_match_the_reference(golden, case):
    from embodied_object_detection_amd import solver
    c = golden["cases"][case]
    s = c["solver"]
    groups = solver.build_param_groups(_named(golden), s["BASE_LR"], s["WEIGHT_DECAY"], s["OPTIMIZER"], s["BACKBONE_MULTIPLIER"],
                                       s["CUSTOM_MULTIPLIER"], s["CUSTOM_MULTIPLIER_NAME"])
    assert [g["name"] for g in groups] == [g["name"] for g in c["groups"]]           # frozen and aliased tensors skipped, order kept
    for g, r in zip(groups, c["groups"]):
        assert g["lr"] == r["lr"], (g["name"], g["lr"], r["lr"])                      # the same float products in the same order
        # ADAMW groups carry no weight_decay of their own: the optimizer's default (the reference passes SOLVER.WEIGHT_DECAY) applies
        assert g.get("weight_decay", c["defaults"]["weight_decay"]) == r["weight_decay"]
        assert ("weight_decay" in g) == (s["OPTIMIZER"] != "ADAMW")


def test_unknown_optimizer_is_refused():
    from embodied_object_detection_amd import solver
    with pytest.raises(NotImplementedError):
        solver.build_param_groups([("a", _P())], 0.1, 0.0, "LAMB")


def test_warmup_cosine_schedule():
    """detectron2's `build_lr_scheduler` for WarmupCosineLR (LRMultiplier over WarmupParamScheduler(CosineParamScheduler(1, 0), ...))
    with the recurrent yaml's numbers (MAX_ITER 10000, WARMUP_ITERS 1000, WARMUP_FACTOR 0.001): inside the warmup a LINE from
    warmup_factor to the cosine's value at the warmup's end (not the older class's product of the two)."""
    from embodied_object_detection_amd.solver import warmup_cosine_lr_factor as f
    cos = lambda w: 0.5 * (1 + math.cos(math.pi * w))
    assert f(0, 10000, 1000, 0.001) == pytest.approx(0.001)
    assert f(500, 10000, 1000, 0.001) == pytest.approx(0.5 * 0.001 + 0.5 * cos(0.1))
    assert f(1000, 10000, 1000, 0.001) == pytest.approx(cos(0.1))
    assert f(6000, 10000, 1000, 0.001) == pytest.approx(cos(0.6))
    assert f(10000, 10000, 1000, 0.001) == pytest.approx(0.0, abs=1e-12)
    assert f(3, 10, 5, 0.2, "constant") == pytest.approx(0.2)
    assert f(7, 10, 5, 0.2, "constant") == pytest.approx(cos(0.7))
    assert f(3, 4, 100, 0.5) == pytest.approx(0.5 * 0.25 + cos(1.0) * 0.75)        # warmup longer than the run: clipped at where = 1


@pytest.mark.gpu
def test_adamw_step_matches_the_reference_optimizer(golden):
    """Two steps of the reference's optimizer (AdamW, per-parameter clip by value 1.0, map_merge at 10 x the base rate) on seeded
    gradients with |g| up to ~9: `eod_adamw_step` reproduces the parameters to fp32 rounding."""
    from embodied_object_detection_amd import ops, solver
    dev = torch.device("cuda:0")
    c = golden["cases"]["recurrent_yaml"]
    s = c["solver"]
    shapes = [tuple(x) for x in golden["shapes"]]
    params = [torch.tensor(v, dtype=torch.float32).reshape(sh).to(dev) for v, sh in zip(c["params_before"], shapes)]
    named = list(zip(golden["names"], params))
    groups = solver.build_param_groups(named, s["BASE_LR"], s["WEIGHT_DECAY"], s["OPTIMIZER"], s["BACKBONE_MULTIPLIER"], s["CUSTOM_MULTIPLIER"],
                                       s["CUSTOM_MULTIPLIER_NAME"], frozen=golden["frozen"])
    opt = ops.AdamW(groups, betas=tuple(c["defaults"]["betas"]), eps=c["defaults"]["eps"], weight_decay=c["defaults"]["weight_decay"],
                    clip_value=s["CLIP_GRADIENTS"]["CLIP_VALUE"])
    by_name = dict(zip(golden["names"], zip(c["grads"], shapes)))
    grads = [torch.tensor(by_name[g["name"]][0], dtype=torch.float32).reshape(by_name[g["name"]][1]).to(dev) for g in groups]
    opt.step(grads)
    opt.step(grads)
    torch.cuda.synchronize()
    for n, p, want, sh in zip(golden["names"], params, c["params_after_two_steps"], shapes):
        ref = torch.tensor(want, dtype=torch.float32).reshape(sh)
        if n in golden["frozen"]:
            assert torch.equal(p.cpu(), ref)
        else:
            assert float((p.cpu() - ref).abs().max()) <= 2e-7 * max(1.0, float(ref.abs().max())), n
            assert not torch.equal(p.cpu(), torch.tensor(c["params_before"][golden["names"].index(n)]).reshape(sh))


@pytest.mark.gpu
def test_multi_tensor_adamw_equals_the_single_tensor_launches():
    """`eod_adamw_step_multi` (all parameter tensors of the training step in ceil(n / 20) launches) against one `eod_adamw_step`
    launch per tensor: bitwise the same parameters and moments after three steps, on 61 tensors of 1 .. 3 M elements with their own
    learning rates, a skipped tensor (no gradient) and clip by value."""
    from embodied_object_detection_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    sizes = [1, 3, 5, 64, 255, 256, 257, 1023, 1024, 1025, 4097, 70001, 3_000_001] + [int(torch.randint(1, 50000, (1,), generator=g)) for _ in range(48)]
    base = [torch.randn((n,), generator=g) for n in sizes]

    folded = {}

    def run(multi):
        params = [b.clone().to(dev) for b in base]
        groups = [{"name": f"t{i}", "param": p, "lr": 1e-3 * (1 + i % 3)} for i, p in enumerate(params)]
        # a trunk conv's raw master [64, 196] whose stepped value x the per-row FrozenBatchNorm scale lands in the layer's padded
        # weights [64, 224] in the same launch (the multi-tensor form) / by a torch expression (the per-tensor form)
        master = torch.linspace(-1, 1, 64 * 196).reshape(64, 196).to(dev)
        layer_w = torch.zeros((64, 224), device=dev)
        scale = (torch.arange(64, dtype=torch.float32) * 0.01 + 0.5).to(dev)
        groups.append({"name": "conv1", "param": master, "lr": 2e-3, "fold": (layer_w, scale)})
        # the same with the gradient of the FOLDED weights handed over (x scale inside the launch: key "grad_of_folded")
        master2 = torch.linspace(1, -1, 64 * 196).reshape(64, 196).to(dev)
        layer_w2 = torch.zeros((64, 224), device=dev)
        groups.append({"name": "conv2", "param": master2, "lr": 2e-3, "fold": (layer_w2, scale), "grad_of_folded": True})
        folded[multi] = (master, layer_w, scale, master2, layer_w2)
        opt = ops.AdamW(groups, weight_decay=1e-2, clip_value=1.0)
        opt.multi_tensor = multi
        gg = torch.Generator().manual_seed(9)
        for step in range(3):
            grads = [(torch.randn((n,), generator=gg) * 3).to(dev) for n in sizes] + [(torch.randn((64, 196), generator=gg)).to(dev)]
            grads.append((grads[-1] / scale.view(-1, 1)) * 0.5)  # conv2: gradient of the folded weights
            if step == 1:
                grads[7] = None                                  # a tensor without a gradient this iteration keeps its step count
            opt.step(grads, lr_factor=0.5 + 0.25 * step)
        torch.cuda.synchronize()
        return params, opt
    pa, oa = run(True)
    pb, ob = run(False)
    assert oa.steps == ob.steps and oa.steps[7] == 2 and oa.steps[0] == 3
    for i in range(len(sizes)):
        assert torch.equal(pa[i], pb[i]), i
        assert torch.equal(oa.state[i][0], ob.state[i][0]) and torch.equal(oa.state[i][1], ob.state[i][1]), i
    assert not torch.equal(pa[12].cpu(), base[12])
    for multi in (True, False):
        master, layer_w, scale, master2, layer_w2 = folded[multi]
        assert torch.equal(layer_w[:, :196], master * scale.view(-1, 1)) and not bool(layer_w[:, 196:].any())
        assert torch.equal(layer_w2[:, :196], master2 * scale.view(-1, 1)) and not bool(layer_w2[:, 196:].any())
    for k in range(5):
        if k != 2:
            assert torch.equal(folded[True][k], folded[False][k]), k
    assert not torch.equal(folded[True][3].cpu(), torch.linspace(1, -1, 64 * 196).reshape(64, 196))
    # the state dict round trip the checkpoint uses
    sd = oa.state_dict()
    oc = ops.AdamW([{"name": f"t{i}", "param": p.clone(), "lr": 1e-3} for i, p in enumerate(pa)] +
                   [{"name": "conv1", "param": folded[True][0].clone(), "lr": 1e-3}, {"name": "conv2", "param": folded[True][3].clone(), "lr": 1e-3}])
    oc.load_state_dict(sd)
    assert oc.steps == oa.steps and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(oc.state, oa.state))
