#!/usr/bin/env python
"""Build an A/B variant of the HIP library with extra compiler flags (diagnostic; never the shipped library):

    python tools/build_variant.py W8 -DWEDM_STAGE_W=8      ->  build/ablate/libwedm_W8.so

Same three-translation-unit parallel build as __graft_entry__.build_hip(); load it with WEDM_HIP_LIB=<path>
(tools/ab_*.sh alternate between the in-tree library and such variants on one box)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

tag, extra = sys.argv[1], sys.argv[2:]
# `--only-part N`: compile translation unit N with the extra flags and take the other units' objects from the in-tree build
# (build/obj/wedm_kernels.part*.o) -- for switches that concern one kernel family only (3 = the served kernels)
only = None
if "--only-part" in extra:
    i = extra.index("--only-part")
    only = int(extra[i + 1])
    extra = extra[:i] + extra[i + 2:]
out = ROOT / "build" / "ablate" / f"libwedm_{tag}.so"
obj_dir = ROOT / "build" / "obj" / tag
obj_dir.mkdir(parents=True, exist_ok=True)
out.parent.mkdir(parents=True, exist_ok=True)
flags = [f for f in g.HIPCC_FLAGS if f != "-shared"] + ["-w", f'-DWEDM_BUILD_ID="{g.kernel_build_id()}+{tag}"'] + extra
procs, objs = [], []
for part in (1, 2, 0, 3, 4):
    if only is not None and part != only:
        objs.append(ROOT / "build" / "obj" / f"wedm_kernels.part{part}.o")
        continue
    obj = obj_dir / f"part{part}.o"
    objs.append(obj)
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, f"-DWEDM_PART={part}", "-c", "-o", str(obj), str(g.HIP_SRC)],
                                  cwd=str(g.HIP_SRC.parent)))
if any(p.wait() != 0 for p in procs):
    raise SystemExit("compile failed")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)], check=True)
print(out)
