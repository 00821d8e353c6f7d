"""Pin oracle/net_oracle.py (torch fp32 functional restatement) to the reference modules' outputs."""
import numpy as np
import pytest
import torch

from oracle import detrand, net_oracle


@pytest.mark.parametrize("bname", ["darknet_21", "darknet_53"])
def test_net_oracle_matches_reference(golden, bname):
    g = golden("g8_network")
    wseed, xseed, cseed, px, bs = [int(v) for v in g[bname + "_meta"]]
    keys = [k for k, _ in net_oracle.state_keys(bname)]
    assert keys == [str(k) for k in g[bname + "_keys"]]        # same state_dict keys, same order
    sd = net_oracle.det_state(bname, wseed)
    for k, v in sd.items():
        if v.dtype == torch.float32:
            v.requires_grad_(not k.endswith(("running_mean", "running_var")))
    x = torch.from_numpy(detrand.uniform(xseed, (bs, 3, px, px), -2.0, 2.0)).requires_grad_(True)
    outs = net_oracle.forward(sd, x, bname, training=True)
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f"{bname}_out{k}"], rtol=1e-3, atol=1e-4)
    cots = [detrand.uniform(cseed + k, tuple(o.shape), -1.0, 1.0) for k, o in enumerate(outs)]
    sum((o * torch.from_numpy(c)).sum() for o, c in zip(outs, cots)).backward()
    np.testing.assert_allclose(x.grad.numpy(), g[bname + "_xgrad"], rtol=1e-2, atol=1e-3 * np.abs(g[bname + "_xgrad"]).max())
    names = [str(n) for n in g[bname + "_pnames"]]
    gn = np.array([float(sd[n].grad.double().norm()) for n in names])
    np.testing.assert_allclose(gn, g[bname + "_gradnorm"], rtol=1e-2)
    # eval mode (running stats = init values in the fixture's eval pass after ONE train step update)
    # the reference ran eval after a train forward, so its running stats were updated once: reproduce
    sd2 = net_oracle.det_state(bname, wseed)
    with torch.no_grad():
        _train_update_running_stats(sd2, torch.from_numpy(detrand.uniform(xseed, (bs, 3, px, px), -2.0, 2.0)), bname)
        eo = net_oracle.forward(sd2, torch.from_numpy(detrand.uniform(xseed, (bs, 3, px, px), -2.0, 2.0)), bname, training=False)
    for k, o in enumerate(eo):
        np.testing.assert_allclose(o.numpy(), g[f"{bname}_evalout{k}"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(sd2["backbone.bn1.running_mean"].numpy(), g[bname + "_rm_stem"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd2["backbone.bn1.running_var"].numpy(), g[bname + "_rv_stem"], rtol=1e-4, atol=1e-6)


def _train_update_running_stats(sd, x, bname):
    """One train-mode forward that updates the running statistics in place (momentum 0.1)."""
    import torch.nn.functional as F
    orig = F.batch_norm

    def bn(inp, rm, rv, w, b, training, mom, eps):
        return orig(inp, rm, rv, w, b, training, mom, eps)
    # re-run forward with running buffers passed in: patch the oracle's batch_norm call
    def patched(inp, rm, rv, w, b, training, mom, eps):
        if training and rm is None:
            # find the buffers belonging to this weight tensor
            for k, v in sd.items():
                if v is w:
                    base = k[: -len(".weight")]
                    return orig(inp, sd[base + ".running_mean"], sd[base + ".running_var"], w, b, True, mom, eps)
        return orig(inp, rm, rv, w, b, training, mom, eps)
    F.batch_norm = patched
    try:
        net_oracle.forward(sd, x, bname, training=True)
    finally:
        F.batch_norm = orig
