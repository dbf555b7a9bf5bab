#!/usr/bin/env python3
"""Can the host store into device memory directly (large BAR)?  hipExtMallocWithFlags(hipDeviceMallocFinegrained) / plain hipMalloc /
hipMallocSignalMemory: write from the CPU in a child process (a fault kills only the child), read back with hipMemcpy."""
import ctypes as C
import subprocess
import sys

CHILD = r'''
import ctypes as C, sys
hip = C.CDLL("libamdhip64.so")
kind = sys.argv[1]
p = C.c_void_p()
if kind == "fine":
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(0x1))
elif kind == "signal":
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(8), C.c_uint(0x2))
elif kind == "uncached":
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(0x3))
else:
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(4096))
print(kind, "alloc rc", rc, hex(p.value or 0), flush=True)
arr = (C.c_uint * 2).from_address(p.value)
arr[0] = 0xC0FFEE
arr[1] = 0x1234
print(kind, "host stores done", flush=True)
out = (C.c_uint * 2)()
rc = hip.hipMemcpy(out, p, C.c_size_t(8), C.c_int(2))
print(kind, "read back rc", rc, hex(out[0]), hex(out[1]), "host load", hex(arr[0]), flush=True)
'''
for kind in ("fine", "signal", "uncached", "plain"):
    r = subprocess.run([sys.executable, "-c", CHILD, kind], capture_output=True, text=True, timeout=120)
    print(f"--- {kind}: exit {r.returncode}\n{r.stdout}{r.stderr[-300:]}")
