#!/usr/bin/env python3
"""Generate golden vectors for the network forward by running the REAL reference
module (azchess/model/resnet.py) in the build container.

Runs only where /root/reference exists (never on the GPU box).  `resnet.py:9` does
`import chess` but never uses it, so an empty stub module is registered under that
name and the file is loaded directly (azchess/__init__.py is skipped).

Output: tests/golden/net_<name>.npz with
    cfg_json, x, p, v, ssl_<task>..., sd::<state-dict key>...
All parameters (including norm gains, rel_bias, biases that the reference
initialises to 0/1) are re-randomised under a fixed seed so every term of the
forward is exercised.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("M0_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference_resnet():
    sys.modules.setdefault("chess", types.ModuleType("chess"))
    spec = importlib.util.spec_from_file_location("ref_resnet", os.path.join(REF, "azchess/model/resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_resnet"] = mod
    spec.loader.exec_module(mod)
    return mod


CASES = {
    # name: (cfg, batch)
    "gn_silu_preact": (dict(planes=19, channels=32, blocks=3, attention_heads=2, policy_size=4672,
                            norm="group", activation="silu", preact=True, policy_factor_rank=16,
                            self_supervised=True,
                            ssl_tasks=["piece", "threat", "pin", "fork", "control"]), 3),
    "gn_dense_leaky": (dict(planes=19, channels=32, blocks=3, attention_heads=2, policy_size=4672,
                            norm="group", activation="silu", preact=True, policy_factor_rank=0,
                            value_activation="leaky_relu", attention_relbias=False,
                            self_supervised=False), 2),
    "bn_relu_postact": (dict(planes=19, channels=32, blocks=2, attention_heads=2, policy_size=4672,
                             norm="batch", activation="relu", preact=False, se=False,
                             attention_every_k=2, policy_factor_rank=8, self_supervised=True,
                             ssl_tasks=["piece"]), 2),
    "stride2": (dict(planes=19, channels=64, blocks=6, attention_heads=4, policy_size=4672,
                     norm="group", activation="silu", preact=True, policy_factor_rank=16,
                     infer_attention_stride=2, value_activation="leaky_relu",
                     self_supervised=True, ssl_tasks=["piece", "control"]), 2),
}


# Full-width cases: the weights are NOT stored.  They are `matrix0_amd.weights.random_state_dict(cfg, seed, varied=True)` (a
# deterministic function of cfg and seed) loaded into the reference's own PolicyValueNet; the fixture holds cfg, seed, x and what
# the reference module computed.  These pin the 320-channel kernels (conv_zs_kernel with every epilogue, attn_block_kernel,
# the big-tile 1x1 convs) to the reference module itself rather than through the oracle.
SEEDED_CASES = {
    # blocks=7, attention_every_k=3: res, res, res+ATTENTION, res, res, res+ATTENTION, res -- one attention block followed by a
    # residual block (the fused block writes the next block's pre-activated input too) and ... the same again, then the heads
    "r320x7_seeded": (dict(planes=19, channels=320, blocks=7, attention_heads=20, policy_size=4672, norm="group",
                           activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
                           ssl_tasks=["piece", "threat", "pin", "fork", "control"]), 21, 8),
}


def gen_seeded(mod):
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if root not in sys.path:
        sys.path.insert(0, root)
    from matrix0_amd.weights import random_state_dict
    for name, (cfg, seed, B) in SEEDED_CASES.items():
        torch.manual_seed(1234)
        net = mod.PolicyValueNet(mod.NetConfig(**cfg)).eval()
        sd = random_state_dict(cfg, seed=seed, varied=True)
        res = net.load_state_dict(sd, strict=False)
        # the only keys of the module the generator does not produce are the aliases of ssl_heads.piece.* (resnet.py:436-437)
        missing = [k for k in res.missing_keys if not (k.startswith("ssl_head.") or k.startswith("ssl_piece_head."))]
        assert not missing and not res.unexpected_keys, (missing, res.unexpected_keys)
        own = dict(net.state_dict())
        for k, t in sd.items():
            assert torch.equal(own[k], t.reshape(own[k].shape).to(own[k].dtype)), k
        g = torch.Generator().manual_seed(1000 + seed)
        x = torch.zeros(B, 19, 8, 8)
        x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
        x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
        x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
        with torch.no_grad():
            p, v, ssl = net(x, return_ssl=True)
        blob = {"cfg_json": np.array(json.dumps(cfg)), "seed": np.array(seed), "x": x.numpy(), "p": p.numpy(), "v": v.numpy()}
        for t, arr in ssl.items():
            blob[f"ssl_{t}"] = arr.numpy()
        path = os.path.join(OUT, f"net_{name}.npz")
        np.savez_compressed(path, **blob)
        print(name, "params", sum(int(t.numel()) for t in sd.values()), "p", tuple(p.shape), "|p|max", float(p.abs().max()),
              "v", v.numpy().round(4), os.path.getsize(path) // 1024, "KiB")


def main():
    mod = load_reference_resnet()
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "seeded":
        gen_seeded(mod)
        return
    for name, (cfg, B) in CASES.items():
        torch.manual_seed(1234)
        net = mod.PolicyValueNet(mod.NetConfig(**cfg)).eval()
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for k, p in net.named_parameters():
                if k == "_policy_logit_scale_raw":
                    continue
                if p.dim() == 1 and k.endswith(".weight"):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                elif k.endswith(".bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                elif k.endswith("rel_bias"):
                    p.copy_(0.7 * torch.randn(p.shape, generator=g))
            # big tensors: snap to fp16-representable values (exact in fp32) so the
            # fixture compresses; the dense policy matrix is regenerated from a seed.
            for k, p in net.named_parameters():
                if p.numel() >= 4096:
                    p.copy_(p.half().float())
                if k == "policy_fc.weight":
                    gg = torch.Generator().manual_seed(4242)
                    p.copy_((0.02 * torch.randn(p.shape, generator=gg)).half().float())
            for k, b in net.named_buffers():
                if k.endswith("running_mean"):
                    b.copy_(0.2 * torch.randn(b.shape, generator=g))
                elif k.endswith("running_var"):
                    b.copy_(0.5 + torch.rand(b.shape, generator=g))
        x = torch.zeros(B, 19, 8, 8)
        # plausible planes: sparse 0/1 piece planes + constant planes
        x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
        x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
        x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
        with torch.no_grad():
            if cfg.get("self_supervised", True):
                p, v, ssl = net(x, return_ssl=True)
            else:  # reference raises AttributeError on return_ssl without SSL heads
                p, v = net(x)
                ssl = None
        blob = {"cfg_json": np.array(json.dumps(cfg)), "x": x.numpy(), "p": p.numpy(), "v": v.numpy()}
        if isinstance(ssl, dict):
            for t, arr in ssl.items():
                blob[f"ssl_{t}"] = arr.numpy()
        seen = set()
        for k, t in net.state_dict().items():
            # ssl_head.* / ssl_piece_head.* are aliases of ssl_heads.piece.* (resnet.py:436-437)
            if k.startswith("ssl_head.") or k.startswith("ssl_piece_head."):
                continue
            if k.endswith("num_batches_tracked"):
                continue
            seen.add(k)
            if k == "policy_fc.weight":
                blob["sdseed::" + k] = np.array([4242, 0.02])  # regenerate: 0.02*randn(seed).half().float()
                continue
            blob["sd::" + k] = t.numpy().astype(np.float32)
        path = os.path.join(OUT, f"net_{name}.npz")
        np.savez_compressed(path, **blob)
        print(name, "params", sum(int(np.prod(blob['sd::' + k].shape)) for k in seen if 'sd::'+k in blob),
              "p", p.shape, "v", v.numpy(), os.path.getsize(path) // 1024, "KiB")


    gen_seeded(mod)


if __name__ == "__main__":
    main()
