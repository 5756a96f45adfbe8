"""Do two commits compile a source file to the same device code? Usage: isa_diff.py <old-commit> <new-commit|WORK> file.hip [...]
Exports both trees to a scratch directory, compiles the named sources device-only to gfx950 assembly and compares the
instruction stream of every kernel (labels normalised). Used at the end of round 3 to show that code added after the last GPU run
(new kernels, host branches) left the instruction streams of the kernels that run had validated untouched."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def export(commit, dst):
    os.makedirs(dst)
    if commit == "WORK":
        subprocess.check_call(f"cp -r {ROOT}/sclens_amd {ROOT}/include {dst}/", shell=True)
    else:
        subprocess.check_call(f"git -C {ROOT} archive {commit} sclens_amd/csrc include | tar -x -C {dst}", shell=True)


def kernels(path):
    out, cur, buf = {}, None, []
    for line in open(path).read().split("\n"):
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if cur:
                out[cur] = "\n".join(buf)
            cur, buf = m.group(1), []
        elif line.startswith(".Lfunc_end"):
            if cur:
                out[cur] = "\n".join(buf)
            cur, buf = None, []
        elif cur is not None:
            body = line.split(";")[0].rstrip()
            if body.strip() and (not body.strip().startswith(".") or body.strip().startswith(".LBB")):
                buf.append(re.sub(r"\.LBB\d+_", ".LBBx_", body))
    return out


def main():
    old, new, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    tmp = tempfile.mkdtemp(prefix="isa_diff_")
    bad = 0
    for tag, commit in (("old", old), ("new", new)):
        export(commit, os.path.join(tmp, tag))
    for f in files:
        asm = {}
        for tag in ("old", "new"):
            d = os.path.join(tmp, tag, "sclens_amd", "csrc")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                                   f, "-o", f + ".s"], cwd=d, stderr=subprocess.DEVNULL)
            asm[tag] = kernels(os.path.join(d, f + ".s"))
        same = [k for k in asm["old"] if k in asm["new"] and asm["old"][k] == asm["new"][k]]
        diff = [k for k in asm["old"] if k in asm["new"] and asm["old"][k] != asm["new"][k]]
        gone = [k for k in asm["old"] if k not in asm["new"]]
        added = [k for k in asm["new"] if k not in asm["old"]]
        print(f"{f}: {len(same)} functions identical, {len(diff)} changed, {len(gone)} removed, {len(added)} added")
        for k in diff:
            print("   changed:", k)
        for k in added:
            print("   added:  ", k)
        bad += len(diff)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
