"""Diagnostic: encoder causality (bitwise) at small and large sizes; per-latent-frame differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vae_oracle as V
from versecrafter_amd.models.wan_vae import AutoencoderKLWan

def make(dim):
    cfg = V.Config(dim=dim, z_dim=16)
    W = {k: v.bfloat16() for k, v in V.random_weights(cfg, 11).items()}
    m = AutoencoderKLWan(latent_channels=16, dim=dim, dim_mult=tuple(cfg.dim_mult), num_res_blocks=cfg.num_res_blocks)
    m.load_state_dict({"model." + k: v for k, v in W.items()})
    return m.to("cuda")

for dim, F, Fh, H, W in ((32, 13, 9, 32, 48), (96, 13, 9, 64, 64), (96, 21, 13, 240, 416), (96, 81, 41, 480, 832), (96, 81, 41, 720, 1280)):
    m = make(dim)
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(1, 3, F, H, W, generator=g) * 2 - 1).bfloat16().cuda()
    lat = m.encode(x)[0].mode()
    head = m.encode(x[:, :, :Fh].contiguous())[0].mode()
    torch.cuda.synchronize()
    n = head.shape[2]
    d = (head.float() - lat[:, :, :n].float())
    per = [float(d[:, :, t].abs().max()) for t in range(n)]
    print(dim, F, Fh, H, W, "equal", torch.equal(head, lat[:, :, :n]), "max|d| per latent frame", [round(p, 4) for p in per],
          "ws GiB", round(m.workspace_bytes() / 2**30, 1), flush=True)
    m.release_workspace()
    del m, x, lat, head
    torch.cuda.empty_cache()
