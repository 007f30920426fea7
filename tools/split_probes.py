#!/usr/bin/env python3
"""Where do the cycles of the split-precision main loop go?  Builds liblshrs_hip.so variants with -DLSHRS_SPLIT_PROBE=n
(each probe removes one ingredient of the loop; keys are wrong by design), runs 1M x 768 through each in a fresh process
and prints the in-kernel stamps: cycles per 16-deep stage (48 MFMAs per wave, floor 1536), workgroup time, shader clock.
Needs hipcc on the GPU box (same image).  usage: split_probes.py [pipe=3|4] [probe ...]"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "lshrs_amd/csrc/liblshrs_hip.so")
NAMES = {128: "16x16x32 MFMAs (wrong math)", 0: "baseline", 1: "no fragment reads", 2: "no bf16 split", 4: "no DMA", 8: "no barrier", 16: "no x read-back",
         31: "bare MFMA stream", 64: "no x pieces (pipe 3 only)"}

def child(probe, pipe):
    sys.path.insert(0, ROOT)
    import torch
    from lshrs_amd import LSHHasher, _native
    lib = _native.load()
    lib.lshrs_debug_set_clock_probe.argtypes = [ctypes.c_void_p]
    lib.lshrs_debug_set_split_pipe(pipe)
    n = 1_000_000
    x = torch.randn(n, 768, device="cuda", generator=torch.Generator("cuda").manual_seed(47))
    out = torch.empty((n, 16, 2), dtype=torch.uint8, device="cuda")
    h = LSHHasher(16, 16, 768, seed=42); h.pipeline_chunk_rows = 10**9; h._flag_cap_hint = 100_000_000
    blocks = (n + 255) // 256
    for _ in range(4): h.hash_device(x, out=out, tie_break="none")
    stamps = torch.zeros(4 * blocks, dtype=torch.int64, device="cuda")
    h.kernel_events = []
    lib.lshrs_debug_set_clock_probe(stamps.data_ptr())
    h.hash_device(x, out=out, tie_break="none"); torch.cuda.synchronize()
    lib.lshrs_debug_set_clock_probe(None)
    ev = h.kernel_events[0]
    loop = stamps[:2 * blocks].view(-1, 2).double().cpu(); whole = stamps[2 * blocks:].view(-1, 2).double().cpu()
    print(f"probe {probe:3d} ({NAMES.get(probe, '?'):26s}) pipe {pipe}: stage-1 kernel {ev[0].elapsed_time(ev[3]):.3f} ms; main loop "
          f"{loop[:, 0].median().item() / 48:.0f} cycles/stage ({loop[:, 1].median().item() / 100:.1f} us), workgroup "
          f"{whole[:, 1].median().item() / 100:.1f} us, shader clock {(loop[:, 0] / loop[:, 1] * 0.1).median().item():.2f} GHz")

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    child(int(sys.argv[2]), int(sys.argv[3]))
    sys.exit(0)
pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 4
probes = [int(a) for a in sys.argv[2:]] or [0, 2, 4, 8, 31]
keep = tempfile.mkdtemp()
shutil.copy(MAIN, os.path.join(keep, "main.so"))
try:
    for p in probes:
        if p:
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                            "-I" + os.path.join(ROOT, "include"), f"-DLSHRS_SPLIT_PROBE={p}",
                            os.path.join(ROOT, "lshrs_amd/csrc/lshrs_hip.hip"),
                            os.path.join(ROOT, "lshrs_amd/csrc/pipeline.hip"), "-o", MAIN], check=True)
        else:
            shutil.copy(os.path.join(keep, "main.so"), MAIN)
        r = subprocess.run([sys.executable, __file__, "--child", str(p), str(pipe)], capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-600:], flush=True)
finally:
    shutil.copy(os.path.join(keep, "main.so"), MAIN)
