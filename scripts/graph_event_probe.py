#!/usr/bin/env python
"""Can HIP events be recorded as event-record NODES of a captured graph on this runtime (hipEventRecordWithFlags(...,
hipEventRecordExternal) through ctypes - torch refuses `Event(external=True)` on ROCm) and timed after a replay?"""
import ctypes
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]


def ev():
    e = ctypes.c_void_p()
    assert hip.hipEventCreate(ctypes.byref(e)) == 0
    return e


a = torch.randn(4096, 4096, device="cuda")
b = torch.randn(512, 512, device="cuda")
evs = [ev() for _ in range(4)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        a @ a; b @ b
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
rc = []
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    st = torch.cuda.current_stream().cuda_stream
    rc.append(hip.hipEventRecordWithFlags(evs[0], st, 1))
    c = a @ a
    rc.append(hip.hipEventRecordWithFlags(evs[1], st, 1))
    d = b @ b
    rc.append(hip.hipEventRecordWithFlags(evs[2], st, 1))
    e = a @ a
    rc.append(hip.hipEventRecordWithFlags(evs[3], st, 1))
print("record rc during capture:", rc)
for it in range(3):
    g.replay()
    torch.cuda.synchronize()
    ms = []
    for i in range(3):
        t = ctypes.c_float()
        r = hip.hipEventElapsedTime(ctypes.byref(t), evs[i], evs[i + 1])
        ms.append((r, round(t.value * 1e3, 1)))
    print("replay", it, "(rc, us) big / small / big matmul:", ms)
# reference: eager events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); a @ a; e1.record(); torch.cuda.synchronize()
print("eager big matmul us:", round(e0.elapsed_time(e1) * 1e3, 1))
