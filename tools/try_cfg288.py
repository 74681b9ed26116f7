import sys, os, json, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import net_ref
from matrix0_amd.backend import M0Backend
cfg = dict(planes=19, channels=288, blocks=22, attention=True, attention_heads=18, policy_size=4672, norm="group",
           activation="silu", value_activation="leaky_relu", preact=True, policy_factor_rank=160,
           infer_attention_stride=2, self_supervised=True, ssl_tasks=["piece", "threat", "pin", "fork", "control"])
sd = net_ref.random_state_dict(cfg, seed=0)
be = M0Backend.from_state_dict(cfg, sd)
print("params", be.param_count(), "gflop", be.flops_per_position(False)/1e9)
g = torch.Generator().manual_seed(5)
B = 6
x = torch.zeros(B, 19, 8, 8)
x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
p_ref, v_ref = net_ref.forward(sd, cfg, x, return_ssl=False)[:2]
p, v = be.infer_np(x.numpy())
print("dlogit", float(np.abs(p - p_ref.numpy()).max()), "dv", float(np.abs(v - v_ref.numpy()).max()))
for Bb in (1024, 4096):
    ms = be.bench_forward(Bb, 3)
    print(json.dumps({"B": Bb, "ms": round(ms, 2), "pos_per_s": round(Bb / ms * 1e3)}))
