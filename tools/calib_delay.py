#!/usr/bin/env python3
"""Check the what-if transport's delay kernel against HIP events (GPU box): vc_sp_init_sim holds a stream for bytes / gbps."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from versecrafter_amd import _lib
from versecrafter_amd.models import VerseCrafterWanTransformer3DModel

m = VerseCrafterWanTransformer3DModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=64, geoada_in_dim=128,
                                      param_device="cuda", param_dtype=torch.bfloat16)
lib, h = _lib.load(), m._engine_handle()
for gbps, nbytes in ((100.0, 100 << 20), (857.0, 220 << 20), (122.0, 503 << 20)):
    _lib.check(lib.vc_sp_init_sim(h, 2, 0, C.c_double(gbps)), h)
    send = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(2 * nbytes, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib.vc_sp_all_to_all(h, 0, send.data_ptr(), recv.data_ptr(), nbytes, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        _lib.check(lib.vc_sp_all_to_all(h, 0, send.data_ptr(), recv.data_ptr(), nbytes, st), h)
    b.record()
    torch.cuda.synchronize()
    want = nbytes / (gbps * 1e9) * 1e3
    print(f"a2a of {nbytes >> 20} MiB/peer at {gbps} GB/s: modelled wire {want:.3f} ms, measured {a.elapsed_time(b) / 5:.3f} ms "
          f"(includes the local copy of {2 * nbytes >> 20} MiB)")
