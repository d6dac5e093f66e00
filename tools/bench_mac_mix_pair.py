#!/usr/bin/env python3
"""Server::mix's point butterflies (Server.hpp:1281-1318) on device-resident arrays: the MAC commitments alone
(porla_icc_mac_mix_device), and commitments + alignments in one launch (porla_icc_mac_mix_pair_device) against two single calls."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from porla_amd import lib
from tests import common
vp = ctypes.c_void_p
n_total = 1 << 17
pool = common.synth_points(4096, start=300)
for lg in (4, 8, 10, 12, 13, 14, 15):
    ln = 1 << lg
    arr = (pool * (4 * ln // 4096 + 1))[:64 * 4 * ln]
    d = [torch.frombuffer(bytearray(arr[64 * ln * k:64 * ln * (k + 1)]), dtype=torch.uint8).cuda() for k in range(4)]
    oa, ob = (torch.empty(128 * ln, dtype=torch.uint8, device="cuda") for _ in range(2))
    s = torch.cuda.current_stream().cuda_stream
    single = lambda a0, a1, o: lib.porla_icc_mac_mix_device(vp(a0.data_ptr()), vp(a1.data_ptr()), ln, n_total, 0, vp(o.data_ptr()), vp(s))
    def two():
        assert single(d[0], d[1], oa) == 0 and single(d[2], d[3], ob) == 0
    def pair():
        assert lib.porla_icc_mac_mix_pair_device(vp(d[0].data_ptr()), vp(d[1].data_ptr()), vp(d[2].data_ptr()), vp(d[3].data_ptr()), ln, n_total, 0,
                                                 vp(oa.data_ptr()), vp(ob.data_ptr()), vp(s)) == 0
    res = {"len": ln}
    two(); torch.cuda.synchronize()
    want = (bytes(oa.cpu().numpy()), bytes(ob.cpu().numpy()))
    pair(); torch.cuda.synchronize()
    res["pair_equals_two_calls"] = (bytes(oa.cpu().numpy()), bytes(ob.cpu().numpy())) == want
    for name, fn in (("two_calls_ms", two), ("pair_call_ms", pair)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    print(json.dumps(res), flush=True)

# ---- the whole Server::mix in one call (porla_server_mix_device): data rows of 128 symbols + both point arrays
for lg in (8, 12, 14):
    ln = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(lg)
    blocks = []
    for _ in range(2):
        t = torch.randint(0, 256, (ln, 128, 64), dtype=torch.uint8, device="cuda", generator=g)
        t[:, :, 63] &= 0x1f
        blocks.append(t)
    arr = (pool * (4 * ln // 4096 + 1))[:64 * 4 * ln]
    d = [torch.frombuffer(bytearray(arr[64 * ln * k:64 * ln * (k + 1)]), dtype=torch.uint8).cuda() for k in range(4)]
    o_data = torch.empty((2 * ln, 128, 64), dtype=torch.uint8, device="cuda")
    oa, ob = (torch.empty(128 * ln, dtype=torch.uint8, device="cuda") for _ in range(2))
    s = torch.cuda.current_stream().cuda_stream
    def one_call():
        assert lib.porla_server_mix_device(vp(blocks[0].data_ptr()), vp(blocks[1].data_ptr()), vp(d[0].data_ptr()), vp(d[1].data_ptr()),
                                           vp(d[2].data_ptr()), vp(d[3].data_ptr()), ln, 128, n_total, 0, vp(o_data.data_ptr()), vp(oa.data_ptr()),
                                           vp(ob.data_ptr()), vp(s)) == 0
    def three_calls():
        assert lib.porla_icc_mix_device(vp(blocks[0].data_ptr()), vp(blocks[1].data_ptr()), ln, 128, n_total, 0, vp(o_data.data_ptr()), vp(s)) == 0
        assert lib.porla_icc_mac_mix_device(vp(d[0].data_ptr()), vp(d[1].data_ptr()), ln, n_total, 0, vp(oa.data_ptr()), vp(s)) == 0
        assert lib.porla_icc_mac_mix_device(vp(d[2].data_ptr()), vp(d[3].data_ptr()), ln, n_total, 0, vp(ob.data_ptr()), vp(s)) == 0
    res = {"whole_mix_len": ln}
    for name, fn in (("three_calls_ms", three_calls), ("porla_server_mix_device_ms", one_call)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    print(json.dumps(res), flush=True)
