"""Which inter-process hand-over of a device allocation works on this box?  (tools/exp, round 6)"""
import ctypes as C
import os
import subprocess
import sys

hip = C.CDLL("libamdhip64.so")


class Handle(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]


hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
hip.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]


def opener(hexh):
    assert hip.hipSetDevice(0) == 0
    h = Handle()
    C.memmove(C.byref(h), bytes.fromhex(hexh), 64)
    p = C.c_void_p()
    rc = hip.hipIpcOpenMemHandle(C.byref(p), h, 1)
    if rc != 0:
        print("open rc", rc)
        return
    buf = (C.c_uint8 * 16)()
    print("open ok, memcpy rc", hip.hipMemcpy(buf, p, 16, 2), bytes(buf).hex())


def main():
    if sys.argv[1] == "open":
        opener(sys.argv[2])
        return
    print("env HSA_ENABLE_IPC_MODE_LEGACY =", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "ptrace_scope:", open("/proc/sys/kernel/yama/ptrace_scope").read().strip() if os.path.exists("/proc/sys/kernel/yama/ptrace_scope") else "n/a")
    assert hip.hipSetDevice(0) == 0
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), 1 << 20) == 0
    assert hip.hipMemset(p, 0x5A, 1 << 20) == 0
    assert hip.hipDeviceSynchronize() == 0
    h = Handle()
    print("get rc", hip.hipIpcGetMemHandle(C.byref(h), p))
    hexh = bytes(h)[:64].hex()
    for label, pre in (("plain", None), ("PR_SET_PTRACER_ANY", lambda: C.CDLL(None).prctl(0x59616D61, C.c_ulong(-1), 0, 0, 0))):
        if pre:
            print("prctl rc", pre())
        r = subprocess.run([sys.executable, __file__, "open", hexh], capture_output=True, text=True, timeout=120)
        print(label, "->", r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:].replace("\n", " | "))
    # a sibling launched through a shell (not a direct child)
    r = subprocess.run("%s %s open %s" % (sys.executable, __file__, hexh), shell=True, capture_output=True, text=True, timeout=120)
    print("via shell ->", r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:].replace("\n", " | "))


if __name__ == "__main__":
    main()
