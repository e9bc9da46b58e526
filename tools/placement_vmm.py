"""(GPU box) What decides the two levels of the fused launch (2.99 / 3.23 ms at mixed-6x64, B = 65536)?  The jvar buffer (8.9 GB) is
built in different ways -- hipMalloc through torch, and the HIP virtual-memory API with physical handles of different sizes, mapped in
order or shuffled, at different virtual alignments -- each way several times in ONE process, x and res staying where they are; the
fused launch is timed on each.  A way that lands on the fast level every time is what Engine should allocate with.
usage: placement_vmm.py [repeats] [workload] [B]   (build/libvmm_alloc.so: tools/microbench/vmm_alloc.hip, built in the container)"""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gelato_amd import Engine, con_dynamics, pack_x, problem
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
wl = sys.argv[2] if len(sys.argv) > 2 else "mixed-6x64"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
L = C.CDLL(os.path.join(ROOT, "build", "libvmm_alloc.so"))
L.vmm_alloc.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
L.vmm_free.argtypes = [C.c_void_p]
L.vmm_granularity.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
pd, ud, c, xd = problem.make_problem(wl)
E = Engine(con_dynamics.problem_arrays(pd, ud))
X = np.tile(problem.synthetic_batch(pack_x(xd), E.M, 64), (B // 64 + 1, 1))[:B]
s = torch.cuda.current_stream().cuda_stream
dX = torch.from_numpy(X).cuda()
r = torch.empty((B, E.nres), dtype=torch.float64, device="cuda")
g = (C.c_size_t * 2)()
print("granularity rc", L.vmm_granularity(0, g), "min", g[0], "recommended", g[1], flush=True)
nbytes = B * E.V * 8


def burst(jp, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        E.eval_batch_device(B, dX.data_ptr(), r.data_ptr(), jp, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def measure(jp):
    burst(jp, 40)
    return [round(burst(jp, 40), 4) for _ in range(3)]


out = {"workload": wl, "batch": B, "jvar_bytes": nbytes, "granularity": [g[0], g[1]], "ways": {}}


def way_torch(i):
    torch.cuda.empty_cache()
    pad = torch.empty(((1 + 37 * i) << 22,), dtype=torch.float64, device="cuda") if i else None
    j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
    del pad
    ms = measure(j.data_ptr())
    del j
    return ms


def way_vmm(chunk, align=0, shuffle=0, offset=0):
    def f(i):
        h, p = C.c_void_p(), C.c_void_p()
        rc = L.vmm_alloc(0, nbytes + offset, chunk, align, (shuffle + i) if shuffle else 0, C.byref(h), C.byref(p))
        if rc:
            return "error %d" % rc
        ms = measure(p.value + offset)
        L.vmm_free(h)
        return ms
    return f


def way_torch_pad1mb(i):
    """as gelato_amd/placement.py: a pad of a random multiple of 1 MB in front"""
    import random
    rng = random.Random(i)
    torch.cuda.empty_cache()
    pad = torch.empty(rng.randrange(64, 4096) * (1 << 17), dtype=torch.float64, device="cuda") if i else None
    j = torch.empty((B, E.V), dtype=torch.float64, device="cuda")
    del pad
    ms = measure(j.data_ptr())
    ptr = j.data_ptr()
    del j
    return ms + [hex(ptr)]


GB, MB = 1 << 30, 1 << 20
ways = [("torch_pad1MB", way_torch_pad1mb)] + [("vmm_off%dMB" % o, way_vmm(0, GB, 0, o * MB)) for o in (0, 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31)] + [("torch_hipMalloc", way_torch), ("vmm_one_handle", way_vmm(0)), ("vmm_2MB", way_vmm(2 * MB)), ("vmm_64MB", way_vmm(64 * MB)),
        ("vmm_1GB", way_vmm(GB, GB)), ("vmm_2MB_shuffled", way_vmm(2 * MB, 0, 1)), ("vmm_1GB_shuffled", way_vmm(GB, GB, 1)),
        ("vmm_one_handle_off4K", way_vmm(0, 0, 0, 4096)), ("vmm_one_handle_off1M", way_vmm(0, 0, 0, MB))]
if len(sys.argv) > 4:
    ways = [w for w in ways if w[0] in sys.argv[4].split(",")]
for name, f in ways:
    res = []
    for i in range(rep):
        try:
            res.append(f(i))
        except Exception as ex:  # noqa: BLE001
            res.append("exception %s" % str(ex)[:100])
        print(name, i, res[-1], flush=True)
    out["ways"][name] = res
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "placement_vmm_%s.json" % wl), "w"), indent=1)
