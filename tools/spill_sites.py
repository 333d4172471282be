"""Where are a translation unit's scratch (spill) accesses: inside a loop or around it?
    python tools/spill_sites.py attn|gemm|norm ...
Compiles carefree-learn_amd/csrc/<name>.hip to gfx950 assembly and lists, per kernel with scratch instructions, how many of
them sit between a loop header and its backward branch."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    for name in sys.argv[1:] or ["attn"]:
        src = os.path.join(ROOT, "carefree-learn_amd", "csrc", name + ".hip")
        out = f"/tmp/spill_{name}.s"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
        for k in re.split(r"\n(?=_Z[^\n]*:\s*;\s*@)", text):
            m = re.match(r"(_Z\S+):", k)
            if not m:
                continue
            body = k.split("s_endpgm")[0].split("\n")
            scr = [i for i, x in enumerate(body) if re.search(r"scratch_(load|store)", x)]
            if not scr:
                continue
            kname = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            kname = kname.replace("(anonymous namespace)::", "")
            loops = []
            for i, x in enumerate(body):
                lm = re.match(r"^(\.LBB\d+_\d+):", x)
                if not lm:
                    continue
                for j in range(i + 1, len(body)):
                    if re.search(r"s_cbranch\S*\s+" + re.escape(lm.group(1)) + r"\b", body[j]):
                        loops.append((i, j))
            inside = sum(1 for i in scr if any(a < i < b for a, b in loops))
            print(f"{name}: {kname[:96]:96s} scratch instructions {len(scr):3d}, inside a loop {inside}")


if __name__ == "__main__":
    main()
