#!/usr/bin/env python3
"""Memory-operation / wait sequence of a kernel's gfx950 ISA, to spot serialized loads.

    python tools/isa_waits.py hifiles-solver_amd/csrc/fused_hex.hip split_update_kernel '<3, 5, true>'

Compiles the file for the device only (-S), finds the kernels whose demangled name contains every given substring and
prints, per kernel, the run-length-coded sequence of
    L global / buffer load      D buffer load to LDS        S store             r / x LDS read / write
    m MFMA                      s scalar load               |B| barrier         wvN / wkN s_waitcnt vmcnt(N) / lgkmcnt(N)
A pattern such as  `L wv0 L wv0 L wv0`  (a load, a full wait, the next load) is one memory latency per load: request all loads
of a step before the first use (a load placed behind a store, an LDS write or a predicate is issued after it).  `L21 wv..`
is what a staging step should look like."""
import re
import subprocess
import sys
import tempfile


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    src, pats = sys.argv[1], sys.argv[2:]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", f.name, src],
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(f.name).read()
    for name in re.findall(r"^(_Z\w+):", txt, re.M):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not all(p in dem for p in pats):
            continue
        i = txt.index(name + ":")
        j = txt.find(".Lfunc_end", i)
        seq = []
        for line in txt[i:j].split("\n"):
            l = line.strip()
            if not l or l[0] in ";.":
                continue
            if l.startswith(("global_load", "buffer_load", "flat_load")):
                seq.append("D" if " lds" in l else "L")
            elif l.startswith(("global_store", "buffer_store", "flat_store")):
                seq.append("S")
            elif l.startswith("ds_read"):
                seq.append("r")
            elif l.startswith("ds_write"):
                seq.append("x")
            elif "mfma" in l:
                seq.append("m")
            elif l.startswith("s_load"):
                seq.append("s")
            elif l.startswith("s_barrier"):
                seq.append("|B|")
            elif l.startswith("s_waitcnt"):
                t = "w"
                m = re.search(r"vmcnt\((\d+)\)", l)
                if m:
                    t += "v" + m.group(1)
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    t += "k" + m.group(1)
                seq.append(t)
        out = []
        for x in seq:
            if out and out[-1][0] == x:
                out[-1][1] += 1
            else:
                out.append([x, 1])
        print(dem.split("(")[0])
        print("  " + " ".join("%s%s" % (x, n if n > 1 else "") for x, n in out))


if __name__ == "__main__":
    main()
