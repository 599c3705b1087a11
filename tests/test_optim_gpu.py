"""Fused HIP LARS step (ias_lars_step) against the multi-tensor torch formulation of the same update on CPU copies
(tests/test_config_optim_cpu.py pins that one against the single-tensor formula of flash's LARS)."""
import pytest
import torch

from inverse_audio_synthesis_amd.optim import LARS

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(257, 129), (8192,), (3,), (70000,), (64, 1025), (5,), (1,)]
    ps = [torch.randn(s, generator=g) * (0.5 + i) for i, s in enumerate(shapes)]
    ps[5].zero_()                                   # a parameter that is exactly zero: ratio 1, no decay
    return [torch.nn.Parameter(p.clone().to(dev)) for p in ps]


@pytest.mark.parametrize("wd", [1e-2, 0.0])
def test_fused_lars_matches_foreach_formulation(lib, dev, wd):
    hip, ref = _params(dev, 3), _params(torch.device("cpu"), 3)
    o_hip = LARS(hip, lr=0.3, weight_decay=wd)
    o_ref = LARS(ref, lr=0.3, weight_decay=wd)
    for step in range(3):
        for i, (a, b) in enumerate(zip(hip, ref)):
            g = torch.randn(b.shape, generator=torch.Generator().manual_seed(100 * step + i))
            if i == 2:
                g.zero_()                           # a gradient that is exactly zero: that parameter must not move
            b.grad = g
            a.grad = g.to(dev)
        if step == 2:                               # the scheduler changed the learning rate
            o_hip.param_groups[0]["lr"] = o_ref.param_groups[0]["lr"] = 0.05
        o_hip.step()
        o_ref.step()
        assert "_hip_tables" in o_hip.__dict__ and "_hip_tables" not in o_ref.__dict__
        for i, (a, b) in enumerate(zip(hip, ref)):
            tol = 2e-6 * max(1.0, float(b.detach().abs().max()))
            assert (a.detach().cpu() - b.detach()).abs().max().item() <= tol, (step, i)
    assert torch.equal(hip[2].detach().cpu(), _params(torch.device("cpu"), 3)[2].detach()) or wd == 0.0


def test_fused_lars_is_deterministic_and_handles_unaligned_views(lib, dev):
    base = torch.randn(100001, generator=torch.Generator().manual_seed(0)).to(dev)
    outs = []
    for _ in range(2):
        buf = base.clone()
        p = torch.nn.Parameter(buf[1:70002])       # 4-byte aligned only
        q = torch.nn.Parameter(buf[70004:])
        o = LARS([p, q], lr=0.1, weight_decay=1e-3)
        p.grad = torch.ones_like(p) * 0.25
        q.grad = torch.linspace(-1, 1, q.numel(), device=dev)
        o.step()
        torch.cuda.synchronize()
        outs.append(buf.clone())
        assert buf[0] == base[0] and torch.equal(buf[70002:70004], base[70002:70004])   # nothing written outside
    assert torch.equal(outs[0], outs[1])


def test_replayed_lars_reads_each_steps_learning_rate_without_a_sync(lib, dev):
    """A captured LARS step replayed in a loop whose host side runs ahead of the GPU (no synchronisation between steps,
    a new learning rate every step, as Trainer._graph_step does under the warm-up schedule) against the eager loop, bit for
    bit.  The learning rate reaches the captured launches through device scalars fed by an asynchronous copy from pinned
    memory: with ONE staging buffer (round 4) step k's copy could read step k+1's value; the ring of guarded slots
    (optim.LARS._sync_group_hyper) cannot.  The parameters are large enough (64 MB) that every replay takes far longer than
    the host needs to queue the next one, and there are more steps than ring slots."""
    n, steps = 16 << 20, 3 * LARS.HYPER_RING + 2
    g = torch.Generator().manual_seed(5)
    init = torch.randn(n, generator=g)
    grad = (torch.randn(n, generator=g) * 0.1).to(dev)
    lrs = [0.3 / (1 + k) for k in range(steps)]

    def run(replayed):
        p = torch.nn.Parameter(init.clone().to(dev))
        opt = LARS([p], lr=lrs[0], weight_decay=1e-3)
        p.grad = grad.clone()
        graph = None
        if replayed:
            opt.step()                                    # (tables built eagerly once; undone below)
            with torch.no_grad():
                p.copy_(init.to(dev))
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                opt.step()
        for k in range(steps):
            opt.param_groups[0]["lr"] = lrs[k]
            if graph is None:
                opt.step()
            else:
                opt.sync_hyper()
                graph.replay()                            # no synchronisation: the host is many steps ahead of the GPU
        torch.cuda.synchronize()
        return p.detach().clone()

    eager, replay = run(False), run(True)
    assert torch.isfinite(eager).all() and not torch.equal(eager.cpu(), init)
    assert torch.equal(eager, replay)


def test_carried_parameter_norms_give_the_same_bits_and_notice_outside_writes(lib, dev):
    """Round 5: the update pass of the fused LARS step leaves sum p_new^2 per chunk (summed in the norm pass' own order) for
    the next step, whose norm pass then reads only the gradient (ias_lars_step_carry).  Ten steps with carried norms against
    ten steps of the plain three-pass form (carry switched off by invalidating before every step): the same parameters, bit
    for bit; and a write to a parameter through torch between two steps (version counter) makes the next step recompute."""
    def run(carry, poke_at=None):
        ps = _params(dev, 7)
        opt = LARS(ps, lr=0.2, weight_decay=1e-2)
        modes = []
        for step in range(10):
            for i, p in enumerate(ps):
                p.grad = (torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 * step + i)) * 0.3).to(dev)
            if not carry:
                opt.invalidate_carried_norms()
            if step == poke_at:
                with torch.no_grad():
                    ps[0].mul_(1.5)                      # somebody else writes a parameter
            st = opt.__dict__.get("_hip_carry", {}).get(0)
            modes.append(bool(st and st["valid"] and st["versions"] == tuple(p._version for p in ps)))
            opt.step()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in ps], modes

    plain, m0 = run(False)
    carried, m1 = run(True)
    assert m0 == [False] * 10 and m1 == [False] + [True] * 9          # (what the step was about to do: recompute / carry)
    for a, b in zip(plain, carried):
        assert torch.equal(a, b)
    poked_plain, _ = run(False, poke_at=4)
    poked_carried, m2 = run(True, poke_at=4)
    assert m2[4] is False and m2[5] is True
    for a, b in zip(poked_plain, poked_carried):
        assert torch.equal(a, b)
    assert not torch.equal(poked_plain[0], plain[0])
