"""Two scans of the built library's gfx950 code objects (no GPU needed):
    python tools/scan_isa.py [libgiddy_amd/libgiddy_amd.so]

1. SERIALIZED SCALAR LOADS: `s_load ... s_waitcnt lgkmcnt(0) ... s_load` within a couple of dozen instructions, past a kernel's first 60 (its arguments) — a load that was issued
   only after the wait for an earlier one, i.e. a round trip of its own.  Round 5 found the run decoders' chunk prologue that way: a conditional
   `anchors[s + 1]` lands in another basic block, behind the wait for `anchors[s]`.  Some are inherent (a batch piece's record, then what it points to).
2. A uniform select on a STALE SCC: hipcc 7.2 compiled `high = new_high < new_low ? new_low : new_high` (wave-uniform 64-bit values) to `v_cmp_gt_u64 vcc, ...` followed by
   `s_cselect_b32` with SCC last written by the `s_addc_u32` of an address computation (tools/experiments/search_probe.hip).  Signature: an s_cselect whose last SCC writer is an
   arithmetic instruction, with a VOPC compare into VCC in between.
3. BARRIERS OF THE DICTIONARY DECODERS: the staged dictionary's trip through LDS overlaps the first tile's key loads, which until round 6 put its __syncthreads() behind a
   flag at three call sites of the tile loop (different wavefronts could arrive at different s_barrier instructions).  Now stream_tiles calls the hook from one place outside
   its loops; the scan asserts what that gives: every dictionary kernel holds at most THREE s_barrier instructions (the overlapped staging's — for a whole first tile, and for
   a ragged or absent one: a workgroup-uniform choice —, and the plain staging loop's of the other branch; the chunk-wise kernels: two), none of them on a cycle of the
   control-flow graph, none behind a narrowing write of EXEC in its basic block.
4. FULL WAITS INSIDE LOOPS (informative): `s_waitcnt vmcnt(0)` on a cycle of the control-flow graph.  CDNA4 counts vector memory operations in order, and a wait for the rows
   requested a trip ago should leave the newer requests in flight (`vmcnt(n)`); behind a wave-uniform branch around a load or a store the compiler no longer knows the count
   and waits for everything — the next rows' loads AND the stores before.  Round 6 found the plain Dictionary <4,8,2>'s chunk loop that way (61 -> 76 % of the HBM peak once
   whole chunks ran straight-line code).  Listed per kernel family: decompressors with the most such waits.  `--loops` prints the table.
"""
import collections, glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
library = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libgiddy_amd", "libgiddy_amd.so")
SCC_WRITER = re.compile(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor|add|sub|addc|subb|cmp|bitcmp|lshl|lshr|ashr|bfe|abs|min|max|not|wqm|bcnt|ff|flbit|absdiff|cmpk|addk|mulk|quadmask)[a-z0-9_]*\b")
ARITHMETIC = re.compile(r"^s_(add|sub|addc|subb|lshl|lshr|ashr|addk|mulk|bfe|abs|min|max|absdiff)[a-z0-9_]*\b")


def demangled(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()


def family(name):
    return re.sub(r"<.*", "", demangled(name).replace("void ", ""))


serialized, stale = collections.defaultdict(list), collections.defaultdict(list)
listing = collections.defaultdict(list)   # decompressors: (address, instruction, branch target relative to the kernel's symbol or None)
kernels = 0
kernel_start = {}
with tempfile.TemporaryDirectory() as scratch:
    copy = os.path.join(scratch, os.path.basename(library))
    shutil.copy(library, copy)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
    for code_object in sorted(glob.glob(os.path.join(scratch, "*gfx950*"))):
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", code_object], capture_output=True, text=True).stdout
        kernel = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                kernel, index, last_load, waited, last_scc, compared = m.group(1), 0, -99, -99, None, False
                kernel_start[kernel] = int(line.split()[0], 16)
                kernels += 1
                continue
            instruction = line.strip().split("//")[0].strip()
            if kernel is None or not instruction:
                continue
            index += 1
            if "decompress" in kernel or "piece_body" in kernel:
                comment = line.split("//")[1] if "//" in line else ""
                address = re.match(r"\s*([0-9A-Fa-f]+):", comment)
                if address:
                    target = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", comment) if instruction.startswith(("s_cbranch", "s_branch")) else None
                    listing[kernel].append((int(address.group(1), 16), instruction, int(target.group(1), 16) if target else None))
            if instruction.startswith(("s_load", "s_buffer_load")):
                if index > 60 and waited > last_load and index - waited <= 20 and waited - last_load <= 8:
                    serialized[kernel].append(index)
                last_load = index
            elif instruction.startswith("s_waitcnt") and "lgkmcnt(0)" in instruction:
                waited = index
            if instruction.startswith("s_cselect") and last_scc and ARITHMETIC.match(last_scc) and compared:
                stale[kernel].append((index, last_scc, instruction))
            if SCC_WRITER.match(instruction) and not instruction.startswith("s_cselect"):
                last_scc, compared = instruction, False
            elif instruction.startswith("v_cmp") and len(instruction.split()) > 1 and instruction.split()[1].startswith("vcc"):
                compared = True

print("%d kernels in %s" % (kernels, library))
print("1. scalar loads issued behind the wait for an earlier one (kernels by family; occurrences in one of them):")
by_family = collections.Counter(family(k) for k in serialized)
for name, count in by_family.most_common():
    example = next(k for k in serialized if family(k) == name)
    print("   %3d kernels  %2d  %s" % (count, len(serialized[example]), name))
print("2. s_cselect on an SCC last written by arithmetic, a VOPC compare in between: %d kernels" % len(stale))
for k, found in list(stale.items())[:10]:
    print("   %s: %s" % (demangled(k)[:120], found[0]))
def barriers_in_cycles(start, instructions):
    """(number of s_barrier instructions, how many of them lie on a cycle of the kernel's control-flow graph)"""
    addresses = [a for a, _, _ in instructions]
    leaders = {addresses[0]}
    for i, (a, text, target) in enumerate(instructions):
        if text.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            if i + 1 < len(instructions):
                leaders.add(addresses[i + 1])
            if target is not None:
                leaders.add(start + target)
    block_of, successors, current = {}, collections.defaultdict(set), None
    for i, (a, text, target) in enumerate(instructions):
        if a in leaders:
            if current is not None and not instructions[i - 1][1].startswith(("s_branch", "s_endpgm")):
                successors[current].add(a)   # falls through
            current = a
        block_of[a] = current
        if target is not None:
            successors[current].add(start + target)
    def on_a_cycle(block):
        seen, stack = set(), list(successors[block])
        while stack:
            b = stack.pop()
            if b == block:
                return True
            if b not in seen:
                seen.add(b); stack.extend(successors[b])
        return False
    found = [a for a, text, _ in instructions if text.startswith("s_barrier")]
    # ... and none under a narrowed EXEC, as far as a basic block shows it: the last instruction before the barrier (in its block) that writes EXEC, if there is one,
    # is the restoring `s_or_b64 exec, exec, saved` of a finished `if`, not a narrowing s_and_saveexec / s_and / s_andn2 / s_mov into exec
    narrowed = 0
    index_of = {a: i for i, (a, _, _) in enumerate(instructions)}
    for a in found:
        i = index_of[a] - 1
        while i >= 0 and block_of[instructions[i][0]] == block_of[a]:
            text = instructions[i][1]
            writes_exec = re.match(r"^s_[a-z0-9_]+\s+exec\b", text) or text.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec"))
            if writes_exec:
                if not text.startswith("s_or_b64 exec, exec,"):
                    narrowed += 1
                break
            i -= 1
    return len(found), sum(1 for a in found if on_a_cycle(block_of[a])) + narrowed


bad_barriers, barrier_counts = [], collections.Counter()
dictionary_kernels = [k for k in listing if "dictionary" in k and ("decompress" in demangled(k).split("(")[0] or "piece_body" in demangled(k).split("(")[0])]
for k in dictionary_kernels:
    count, in_cycles = barriers_in_cycles(kernel_start[k], listing[k])
    barrier_counts[count] += 1
    if count > 3 or in_cycles:
        bad_barriers.append((k, count, in_cycles))
print("3. dictionary decoders: %d kernels, s_barrier instructions per kernel %s; more than three, or one inside a loop: %d kernels" % (
    len(dictionary_kernels), sorted(barrier_counts.items()), len(bad_barriers)))
for k, count, inside in bad_barriers[:10]:
    print("   %s: %d s_barrier, %d inside a loop" % (demangled(k)[:140], count, inside))
if "--loops" in sys.argv:
    def full_waits_in_cycles(start, instructions):
        addresses = [a for a, _, _ in instructions]
        leaders = {addresses[0]}
        for i, (a, text, target) in enumerate(instructions):
            if text.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                if i + 1 < len(instructions):
                    leaders.add(addresses[i + 1])
                if target is not None:
                    leaders.add(start + target)
        block_of, successors, current = {}, collections.defaultdict(set), None
        for i, (a, text, target) in enumerate(instructions):
            if a in leaders:
                if current is not None and not instructions[i - 1][1].startswith(("s_branch", "s_endpgm")):
                    successors[current].add(a)
                current = a
            block_of[a] = current
            if target is not None:
                successors[current].add(start + target)
        cyclic = {}
        def on_a_cycle(block):
            if block not in cyclic:
                seen, stack, found = set(), list(successors[block]), False
                while stack and not found:
                    b = stack.pop()
                    if b == block: found = True
                    elif b not in seen: seen.add(b); stack.extend(successors[b])
                cyclic[block] = found
            return cyclic[block]
        waits = [a for a, text, _ in instructions if text.startswith("s_waitcnt") and "vmcnt(0)" in text]
        return len(waits), sum(1 for a in waits if on_a_cycle(block_of[a]))
    print("4. s_waitcnt vmcnt(0) on a cycle of the control-flow graph (of all such waits), decompressors, most first:")
    rows = []
    for k in listing:
        if k in kernel_start and listing[k]:
            total, in_loops = full_waits_in_cycles(kernel_start[k], listing[k])
            rows.append((in_loops, total, demangled(k).replace("void cuda::kernels::", "").split("(")[0]))
    for in_loops, total, name in sorted(rows, reverse=True)[:int(os.environ.get("SCAN_TOP", "60"))]:
        print("   %3d of %3d  %s" % (in_loops, total, name[:150]))
sys.exit(1 if stale or bad_barriers else 0)
