"""The C ABI used WITHOUT Python bindings: examples/c_abi_denoise_step.cpp (plain C++ host, hipMalloc'd buffers, linked
against libvcengine.so) runs one denoise-step forward from a flat bundle of weights and inputs; its output must equal the
Python drop-in class's output bit for bit (same library, same kernels) and match the CPU oracle within the usual bound."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import wan_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "c_abi_denoise_step")
TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=4, text_dim=64, text_len=48, geoada_in_dim=128,
            in_dim=16, out_dim=16, freq_dim=256)
KIND = {torch.bfloat16: 0, torch.float32: 1, torch.float64: 2, torch.int32: 3}


def _record(name, t):
    t = t.contiguous().cpu()
    raw = t.view(torch.uint8).numpy().tobytes() if t.dtype == torch.bfloat16 else t.numpy().tobytes()
    nb = name.encode()
    return (struct.pack("<I", len(nb)) + nb + struct.pack("<II", KIND[t.dtype], t.dim()) +
            struct.pack(f"<{t.dim()}q", *t.shape) + struct.pack("<Q", len(raw)) + raw)


def test_cpp_host_through_c_abi_equals_python_path(tmp_path):
    assert os.path.isfile(EXE), f"{EXE} is not built (make -C versecrafter_amd/csrc example, done by __graft_entry__.build())"
    from versecrafter_amd.models import VerseCrafterWanTransformer3DModel
    cfg = O.Config(**TINY)
    W = O.random_weights(cfg, 7)
    g = torch.Generator().manual_seed(2025)
    T, h, w = 3, 8, 12
    x = torch.randn(2, 16, T, h, w, generator=g).bfloat16()
    geo = torch.randn(2, 128, T, h, w, generator=g).bfloat16()
    ctx = [torch.randn(20, 64, generator=g).bfloat16(), torch.randn(33, 64, generator=g).bfloat16()]
    t = torch.tensor([875.0, 875.0])
    seq_len = O.seq_len_for((16, T, h, w))

    model = VerseCrafterWanTransformer3DModel(**TINY)
    model.load_state_dict(W)
    model = model.to(torch.bfloat16).to("cuda")
    want = model(x.cuda(), t.cuda(), geo.cuda(), [c.cuda() for c in ctx], seq_len).cpu()

    eps_bits = struct.unpack("<i", struct.pack("<f", float(model.eps)))[0]
    cfg_rec = torch.tensor([TINY["dim"], TINY["ffn_dim"], TINY["num_heads"], TINY["num_layers"], TINY["in_dim"], TINY["out_dim"],
                            TINY["geoada_in_dim"], TINY["text_dim"], TINY["text_len"], TINY["freq_dim"], eps_bits, 0],
                           dtype=torch.int32)
    recs = [_record("cfg", cfg_rec),
            _record("rope", torch.view_as_real(model.freqs.to(torch.complex128).cpu()).contiguous()),
            _record("x", x), _record("t", t), _record("geoada_context", geo),
            _record("seq_len", torch.tensor([seq_len], dtype=torch.int32))]
    recs += [_record(f"ctx.{i}", c) for i, c in enumerate(ctx)]
    recs += [_record("w." + k, v.detach().to(torch.bfloat16)) for k, v in model.state_dict().items()]
    bundle, out = str(tmp_path / "bundle.bin"), str(tmp_path / "out.bin")
    with open(bundle, "wb") as f:
        f.write(struct.pack("<II", 0x31424356, len(recs)))
        for r in recs:
            f.write(r)

    r = subprocess.run([EXE, bundle, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    print(r.stdout.strip())
    got = torch.from_numpy(np.fromfile(out, dtype=np.uint16).copy()).view(torch.bfloat16).reshape(want.shape)
    assert torch.equal(got, want)
    ref = O.forward({k: v.bfloat16().float() for k, v in W.items()}, cfg, x.float(), t, geo.float(),
                    [c.float() for c in ctx], seq_len)
    rel = ((got.float() - ref).norm() / ref.norm()).item()
    assert rel < 3e-2, rel
