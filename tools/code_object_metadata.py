"""Per-kernel metadata of the gfx950 code objects inside a built library — what hipFuncGetAttributes would answer, read WITHOUT a GPU:
name (demangled), max_flat_workgroup_size (__launch_bounds__), static LDS, vector / scalar registers, spills, scratch.
    python tools/code_object_metadata.py [libgiddy_amd/libgiddy_amd.so] [substring of the kernel names to print]
Used by tests/test_boundary.py::test_every_build_of_a_run_decoder_accepts_the_configuration_resolved_for_the_base_kernel."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("max_flat_workgroup_size", "group_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")


def kernels_of(library):
    """{demangled kernel name: {field: int}} over every gfx950 code object bundled in `library`"""
    out = {}
    with tempfile.TemporaryDirectory() as scratch:
        copy = os.path.join(scratch, os.path.basename(library))
        shutil.copy(library, copy)   # (llvm-objdump --offloading writes the bundles next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
        mangled = {}
        for name in sorted(os.listdir(scratch)):
            if "gfx950" not in name:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(scratch, name)], check=True, capture_output=True, text=True).stdout
            current = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
                if not m:
                    continue
                key, value = m.groups()
                if key == "agpr_count" and current.get("name"):      # (first field of the NEXT kernel's record)
                    mangled[current["name"]] = current
                    current = {}
                if key == "name" and value.startswith("_Z") and "symbol" not in current:
                    current["name"] = value
                elif key in FIELDS:
                    current[key] = int(value)
            if current.get("name"):
                mangled[current["name"]] = current
        names = list(mangled)
        demangled = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        for raw, pretty in zip(names, demangled):
            out[pretty] = {k: v for k, v in mangled[raw].items() if k != "name"}
    return out


if __name__ == "__main__":
    library = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libgiddy_amd", "libgiddy_amd.so")
    wanted = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, record in sorted(kernels_of(library).items()):
        if wanted in name:
            print(name[:150], record)
