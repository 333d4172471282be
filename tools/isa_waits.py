"""Per-kernel summary of the waits / barriers / DMA / LDS reads / MFMAs inside the innermost loops of a gfx950 assembly file.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only carefree-learn_amd/csrc/gemm.hip -o /tmp/g.s
    python tools/isa_waits.py /tmp/g.s [kernel-name-regex]

Round 3 used it to find the compiler-inserted `s_waitcnt vmcnt(0)` in front of the LDS reads of every K loop that issued its
LDS-DMA through the builtin (csrc/common.h: lds_dma16)."""
import re
import subprocess
import sys


def main() -> None:
    text = open(sys.argv[1]).read()
    flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    kernels = re.split(r"\n(?=_Z[^\n]*:\s*;\s*@)", text)
    for k in kernels:
        m = re.match(r"(_Z\S+):", k)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "")
        if flt and not flt.search(name):
            continue
        body = k.split("s_endpgm")[0].split("\n")
        # innermost loops: from a "Loop Header" label to the backward branch to it
        i = 0
        printed = False
        while i < len(body):
            lm = re.match(r"^(\.LBB\d+_\d+):.*Inner Loop Header", body[i])
            if not lm:
                i += 1
                continue
            label = lm.group(1)
            j = i + 1
            while j < len(body) and not re.search(r"s_cbranch\S*\s+" + re.escape(label) + r"\b", body[j]):
                j += 1
            loop = body[i:j + 1]
            if not any("v_mfma" in x for x in loop):
                i = j + 1
                continue
            if not printed:
                print(f"== {name[:150]}")
                printed = True
            seq, run, kind = [], 0, None
            for x in loop:
                x = x.strip()
                k2 = ("mfma" if x.startswith("v_mfma") else "ds_read" if x.startswith("ds_read") else
                      "dma" if ("buffer_load" in x and " lds" in x) else None)
                if k2 is None and not re.match(r"s_waitcnt|s_barrier|s_setprio", x):
                    continue
                if k2 != kind and run:
                    seq.append(f"{run}x{kind}")
                    run = 0
                kind = k2
                if k2:
                    run += 1
                else:
                    seq.append(x.replace("s_waitcnt ", "W:").replace("s_barrier", "BAR").replace("s_setprio ", "prio"))
            if run:
                seq.append(f"{run}x{kind}")
            print("   loop", label, "|", " ".join(seq))
            i = j + 1


if __name__ == "__main__":
    main()
