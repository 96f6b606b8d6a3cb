"""Developer tool: where a Raccess wavefront spends its cycles, phase by phase.
Needs the instrumented build (`make -C priblast_amd/csrc prof` -> libpriblast_hip_prof.so) and a GPU.
usage: raccess_profile.py [nseq] [length]"""
import ctypes
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from priblast_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "priblast_amd", "lib", "libpriblast_hip_prof.so")
NAMES = {1: "inside 1: stem, multi2", 2: "inside 2: multibif, multi1", 3: "inside 3: multi / outer chains", 4: "inside 4: stemend fold",
         8: "outside A: stemend copy", 9: "outside B: multi / outer chains", 10: "outside C: multi1, multibif", 11: "outside D: multi2",
         12: "outside E: stem fold"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rng = random.Random(2)
    seqs = ["".join(rng.choice("ACGU") for _ in range(L)) for _ in range(n)]
    lib = capi.lib()
    buf = (ctypes.c_ulonglong * 32)()
    with capi.Context(0) as ctx:
        ctx.accessibility(seqs[:4], 70, 5)
        lib.prb_debug_ra_profile(buf, 1)
        ctx.reset_timers()
        ctx.accessibility(seqs, 70, 5)
        lib.prb_debug_ra_profile(buf, 1)
        print(f"{n} x {L} nt: device {ctx.stage_ms('raccess')[0]:.1f} ms")
    total = sum(buf[:16])
    for k, name in NAMES.items():
        print(f"  {name:34s} {buf[k] / 1e6:10.2f} Mcycles  {buf[k] / total * 100:5.1f} %   {buf[k] / L:10.0f} cycles per column")




if __name__ == "__main__":
    main()
