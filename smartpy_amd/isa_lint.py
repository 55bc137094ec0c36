"""Checks of the BUILT library's gfx950 code that the sources alone cannot give: the pieces of the kernels whose
correctness rests on where hipcc's assembler put an instruction.

  pair blocks   smart_forcing_scan writes byte offsets ("code words") into the blocks of the streaming step loops
                (smart_fast_arms.h: SMART_A_PAIRS_STRETCH, SMART_A_GAP_STREAM, SMART_A_EVERY_STREAM); a block that
                outgrew its room or changed its place would send a computed jump into the middle of another
  hand-over     the publish / wait sequences of the time-sliced kernels (smart_device.h: publish_slice,
                wait_for_slice) as MI355X_MICROARCH.md prescribes them -- hipcc drops a wait it can prove unnecessary
  rows          the DPP chains of the row form of the literal step (smart_literal_lanes.h): inline asm is opaque to
                hipcc's hazard recogniser, so the two wait states gfx950 wants between a vector write of a register
                and a DPP read of it are the source's own business

Round 4 ran the first two from the test suite only.  Now smartpy_amd.build runs all three on every library it links
(check_library) and writes the outcome next to it (<lib>.lint.json, with `hipcc --version` and the library's
sha256); smartpy_amd._lib reads that file when it loads the library: a library whose pair blocks failed the lint -- or
that is not the one the lint looked at, e.g. rebuilt by another hipcc on another box -- runs the threaded chunks
(SMART_PAIR_BLOCKS=0: the same arithmetic without computed jumps) behind a warning; a library whose hand-over or row
chains failed is refused.  The tests (tests/test_pair_blocks_isa.py, test_handover_isa.py, test_lanes_isa.py) call
the same functions.

llvm-objdump of the ROCm toolchain is needed; without it check_library() says so and nothing is claimed.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import tempfile



def _find_objdump():
    """llvm-objdump of the ROCm toolchain: $ROCM_PATH, the usual place, or whatever the PATH holds (round 5 knew one
    hard-coded path: advisor)."""
    roots = [os.environ.get('ROCM_PATH'), os.environ.get('ROCM_HOME'), '/opt/rocm']
    for root in roots:
        if root:
            exe = os.path.join(root, 'lib', 'llvm', 'bin', 'llvm-objdump')
            if os.path.exists(exe):
                return exe
    return shutil.which('llvm-objdump') or '/opt/rocm/lib/llvm/bin/llvm-objdump'


OBJDUMP = _find_objdump()

STAMP_UNCHECKED = b'SMART_LINT_STAMP=unchecked'
STAMP_PAIRS_OK = b'SMART_LINT_STAMP=pairs-ok\0'          # (same length: written over the other in the file)


def stamp_library(path, pairs_ok):
    """The lint's verdict on the pair blocks INTO the library file (smart_capi.hip: smart_lint_stamp), so that a caller of
    the C ABI who never sees the record next to the file gets the computed jumps only from a library whose code was looked
    at.  -> True if the stamp was found (once) and written."""
    with open(path, 'rb') as fh:
        blob = fh.read()
    assert len(STAMP_PAIRS_OK) == len(STAMP_UNCHECKED)
    if blob.count(STAMP_UNCHECKED) != 1:
        return False
    if pairs_ok:
        with open(path, 'wb') as fh:
            fh.write(blob.replace(STAMP_UNCHECKED, STAMP_PAIRS_OK))
    return True


def library_stamp(path):
    """'pairs-ok' | 'unchecked' | None (no stamp in the file)"""
    with open(path, 'rb') as fh:
        blob = fh.read()
    if STAMP_PAIRS_OK.rstrip(b'\0') in blob:
        return 'pairs-ok'
    return 'unchecked' if STAMP_UNCHECKED in blob else None
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'csrc', 'libsmart_amd.so')
FP64 = ('v_fma_f64', 'v_fmac_f64', 'v_add_f64', 'v_mul_f64', 'v_min_f64', 'v_max_f64', 'v_ldexp_f64')

SLICED = ['smart_fast_intervals_exits', 'smart_fast_intervals', 'smart_fast_intervals_states', 'smart_fast_runs_exits',
          'smart_fast_runs', 'smart_fast_runs_states', 'smart_fast_steps', 'smart_fast_steps_states',
          'smart_fast_steps_raw', 'smart_fast_intervals_raw', 'smart_fast_steps_every']
PAIRED = ['smart_fast_steps', 'smart_fast_steps_raw', 'smart_fast_steps_states']
ROWS = ['smart_fast_illcond', 'smart_ensemble_literal_rows']

KINDS = 'CDR'          # smart_device.h: step_kind -- 0 calm, 1 dry, 2 rain
FIRST = {'C': 'v_cmp_lt_f64', 'D': 'v_mul_f64', 'R': 'v_mov_b64'}      # how the three arms begin


class LintError(Exception):
    pass


def _need(cond, message):
    if not cond:
        raise LintError(message)


def classify(op):
    base = op.replace('_e32', '').replace('_e64', '')
    if base in FP64:
        return 'fp64'
    if base.startswith('v_cmp'):
        return 'vcmp'
    if base.startswith(('v_mov', 'v_cndmask', 'v_accvgpr')):
        return 'vmov'
    if base.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')):
        return 'lane'
    if base.startswith('v_'):
        return 'valu'
    if base.startswith(('s_branch', 's_cbranch')):
        return 'branch'
    if base.startswith(('s_load', 's_store', 's_buffer_load', 's_dcache')):
        return 'smem'
    if base.startswith(('global_', 'buffer_', 'flat_', 'scratch_', 'ds_')):
        return 'vmem'
    if base.startswith(('s_waitcnt', 's_nop', 's_sleep', 's_endpgm', 's_barrier', 's_code_end', 's_setprio', 's_trap')):
        return 'other'
    if base.startswith('s_'):
        return 'salu'
    return 'other'


def parse(start, body):
    insts = []
    for line in body.split('\n'):
        m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):\s*[0-9A-F ]+(?:<[^+>]+\+0x([0-9a-f]+)>)?\s*$', line)
        if not m:
            continue
        op, args, addr, off = m.group(1), m.group(2), int(m.group(3), 16), m.group(4)
        target = start + int(off, 16) if off is not None and classify(op) == 'branch' else None
        insts.append({'addr': addr, 'op': op, 'args': args, 'cls': classify(op), 'target': target})
    return insts


class Disassembly(object):
    """The gfx950 code objects bundled in a shared library, disassembled once (llvm-objdump --offloading writes them
    next to its input: done on a copy in a scratch directory); kernel(name) -> its instructions."""

    def __init__(self, lib=LIB):
        if not os.path.exists(OBJDUMP):
            raise LintError('llvm-objdump of the ROCm toolchain not found (%s)' % OBJDUMP)
        self.lib = lib
        tmp = tempfile.mkdtemp(prefix='smart_isa_')
        try:
            copy = os.path.join(tmp, os.path.basename(lib))
            shutil.copy(lib, copy)
            subprocess.run([OBJDUMP, '--offloading', copy], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            self.texts = [subprocess.run([OBJDUMP, '-d', os.path.join(tmp, f)], check=True, capture_output=True,
                                         text=True).stdout for f in sorted(os.listdir(tmp)) if 'gfx950' in f]
        finally:
            shutil.rmtree(tmp)
        self._cache = {}

    def raw(self, kernel):
        for text in self.texts:
            m = re.search(r'^([0-9a-f]+) <(_ZN5smart\d+%s[A-Z][^>]*)>:\n(.*?)(?=^\S|\Z)' % re.escape(kernel), text,
                          re.M | re.S)
            if m:
                return int(m.group(1), 16), m.group(2), m.group(3)
        raise LintError('kernel %s not found in %s' % (kernel, self.lib))

    def kernel(self, name):
        if name not in self._cache:
            start, _, body = self.raw(name)
            insts = parse(start, body)
            for a, b in zip(insts, insts[1:]):
                a['size'] = b['addr'] - a['addr']
            if insts:
                insts[-1]['size'] = 4
            self._cache[name] = insts
        return self._cache[name]


def _disassembly(lib_or_dis):
    return lib_or_dis if isinstance(lib_or_dis, Disassembly) else Disassembly(lib_or_dis or LIB)


def _define(name):
    text = open(os.path.join(HERE, 'csrc', 'smart_device.h')).read()
    return int(re.search(r'#define %s (\d+)' % name, text).group(1))


def fast_kernel_names():
    """the names the library itself lists (kFastKernelNames in smart_capi.hip)"""
    text = open(os.path.join(HERE, 'csrc', 'smart_capi.hip')).read()
    table = re.search(r'kFastKernelNames\[kNumFastKernels\] = \{(.*?)\};', text, re.S).group(1)
    return re.findall(r'"(smart_fast_\w+)"', table)


# ---- pair blocks ----------------------------------------------------------------------------------------------------
def _block_base(insts, i):
    return insts[i]['addr'] + 4 + int(insts[i + 1]['args'].split(',')[-1], 0)


def _entries(insts):
    return [i for i, x in enumerate(insts) if x['op'] == 's_getpc_b64' and 's[78:79]' in x['args']]


def _check_pair_instance(insts, at, i, stride):
    _need(insts[i + 1]['op'] == 's_add_u32' and insts[i + 1]['args'].startswith('s78, s78,'), 'entry without its s_add_u32')
    base = _block_base(insts, i)
    _need(base % 64 == 0, 'block 0 at %#x is not on a 64-byte line' % base)
    # the entry (the second form has two: a stretch may start in either buffer) ends with the jump to the first block,
    # nothing falls into the blocks
    j, last = i, insts[i]
    while insts[j]['addr'] < base:
        last = insts[j]
        j += 1
    while last['op'] == 's_nop':
        j -= 1
        last = insts[j - 1]
    _need(last['op'] == 's_setpc_b64', 'the entry falls into block 0 (%s in front of it)' % last['op'])

    def block(n, names, tail_loads):
        b = base + n * stride
        _need(b in at, 'block %d does not start on an instruction' % n)
        k = at[b]
        if names[0] == 'R':         # entered 4 bytes in (pair_code adds 4): an s_nop on the boundary
            _need(insts[k]['op'] == 's_nop' and insts[k]['size'] == 4, 'block %d (%s): no s_nop on the boundary' % (n, names))
            k += 1
        _need(insts[k]['op'].startswith(FIRST[names[0]]), 'block %d (%s) begins with %s' % (n, names, insts[k]['op']))
        loads = 0
        while insts[k]['op'] != 's_setpc_b64':          # the main path: up to the computed jump
            _need(insts[k]['addr'] < b + stride, 'block %d (%s) outgrew its %d bytes' % (n, names, stride))
            loads += insts[k]['op'].startswith('s_load_dwordx16')
            k += 1
        _need(insts[k]['args'].strip() == 's[76:77]', 'block %d (%s): jump through %s' % (n, names, insts[k]['args']))
        _need(loads == tail_loads, 'block %d (%s) requests %d chunks' % (n, names, loads))
        # what follows the jump (out-of-line cascades) stays inside the block's room and ends with a branch back
        end = insts[k]['addr'] + 4
        m = k + 1
        while m < len(insts) and insts[m]['addr'] < b + stride:
            if insts[m]['op'].startswith(('v_', 's_branch', 's_cbranch')):
                end = insts[m]['addr'] + insts[m]['size']
            m += 1
        _need(end <= b + stride or n == 39, 'block %d (%s): its out-of-line code outgrew the room' % (n, names))

    n = 0
    for pos in range(4):                    # buffer 0: first pair, second pair; buffer 1: first, second
        for k0 in KINDS:
            for k1 in KINDS:
                block(n, k0 + k1, tail_loads=pos % 2)       # the second pair's tail requests the chunk after next
                n += 1
    for _ in range(2):                      # whole chunks: four calm, four dry steps
        for k0 in 'CD':
            block(n, k0 * 4, tail_loads=1)
            n += 1


def _check_gap_stream(insts, at, i, stride):
    """54 blocks: 2 buffers x 3 variants (no report in the pair / behind its first arm / behind its second) x 9 patterns;
    every main path ends with the computed jump inside the block's room, requests one pair, and holds the report's
    row-pointer move exactly where its variant says"""
    base = _block_base(insts, i)
    _need(base % 64 == 0, 'stream block 0 at %#x is not on a 64-byte line' % base)
    for n in range(54):
        variant, pattern = (n % 27) // 9, n % 9
        names = KINDS[pattern // 3] + KINDS[pattern % 3]
        b = base + n * stride
        _need(b in at, 'stream block %d does not start on an instruction' % n)
        k = at[b]
        if names[0] == 'R':
            _need(insts[k]['op'] == 's_nop' and insts[k]['size'] == 4, 'stream block %d: no s_nop on the boundary' % n)
            k += 1
        _need(insts[k]['op'].startswith(FIRST[names[0]]), 'stream block %d (%s) begins with %s' % (n, names, insts[k]['op']))
        loads = reports = 0
        while insts[k]['op'] != 's_setpc_b64':
            _need(insts[k]['addr'] < b + stride, 'stream block %d (%s) outgrew its %d bytes' % (n, names, stride))
            loads += insts[k]['op'].startswith('s_load_dwordx16')
            reports += insts[k]['op'] == 'v_lshl_add_u64'
            k += 1
        _need(loads == 1 and reports == (variant != 0), 'stream block %d (%s): %d loads, %d reports' % (n, names, loads, reports))


def lint_pair_blocks(lib=None, kernel=None):
    """Every block of the streaming step loops lies where smart_forcing_scan's code words point.  Raises LintError."""
    dis = _disassembly(lib)
    for name in ([kernel] if kernel else PAIRED):
        split = name.endswith('_states')    # the models with the final state vector: larger blocks, no stream of records
        stride = _define('SMART_PS_STRIDE' if split else 'SMART_P_STRIDE')
        _need(stride % 64 == 0, 'stride %d is no multiple of 64' % stride)
        insts = dis.kernel(name)
        at = {x['addr']: i for i, x in enumerate(insts)}
        entries = _entries(insts)
        # the stream of records (SMART_A_GAP_STREAM) loads ONE code word per pair, the pair blocks two per chunk
        pairs = [i for i in entries if not any(x['op'] == 's_load_dword' for x in insts[i:i + 12])]
        _need(len(pairs) == 2, '%s: %d instances of the stretch asm (two: even / any number of chunks)' % (name, len(pairs)))
        try:
            for i in pairs:
                _check_pair_instance(insts, at, i, stride)
            streams = [i for i in entries if i not in pairs]
            _need(len(streams) == (0 if split else 1), '%s: %d streams of records' % (name, len(streams)))
            if streams:
                _check_gap_stream(insts, at, streams[0], _define('SMART_E_STRIDE'))
        except LintError as e:
            raise LintError('%s: %s' % (name, e))
    if kernel is None or kernel == 'smart_fast_steps_every':
        lint_every_stream(dis)


def lint_every_stream(lib=None):
    """SMART_A_EVERY_STREAM (a report every step): four instances in smart_fast_steps_every (matrix stored or not,
    observations or not), each 2 x 9 blocks SMART_E_STRIDE bytes apart; every block's main path -- arm, report, arm,
    report, loop control -- ends with the computed jump inside the block's room and requests exactly one pair of steps"""
    dis = _disassembly(lib)
    stride = _define('SMART_E_STRIDE')
    _need(stride % 64 == 0, 'stride %d is no multiple of 64' % stride)
    insts = dis.kernel('smart_fast_steps_every')
    at = {x['addr']: i for i, x in enumerate(insts)}
    entries = _entries(insts)
    _need(len(entries) == 4, 'smart_fast_steps_every: %d instances of the stream (four)' % len(entries))
    stores = []
    for i in entries:
        base = _block_base(insts, i)
        _need(base % 64 == 0, 'smart_fast_steps_every: block 0 at %#x is not on a 64-byte line' % base)
        n_store = 0
        for n in range(18):
            names = KINDS[(n % 9) // 3] + KINDS[n % 3]
            b = base + n * stride
            _need(b in at, 'smart_fast_steps_every: block %d does not start on an instruction' % n)
            k = at[b]
            if names[0] == 'R':
                _need(insts[k]['op'] == 's_nop' and insts[k]['size'] == 4, 'smart_fast_steps_every: block %d: no s_nop' % n)
                k += 1
            _need(insts[k]['op'].startswith(FIRST[names[0]]),
                  'smart_fast_steps_every: block %d (%s) begins with %s' % (n, names, insts[k]['op']))
            loads = 0
            while insts[k]['op'] != 's_setpc_b64':
                _need(insts[k]['addr'] < b + stride, 'smart_fast_steps_every: block %d (%s) outgrew its %d bytes' % (n, names, stride))
                loads += insts[k]['op'].startswith('s_load_dwordx16')
                n_store += insts[k]['op'] == 'global_store_dwordx2'
                k += 1
            _need(loads == 1, 'smart_fast_steps_every: block %d (%s) requests %d pairs' % (n, names, loads))
        stores.append(n_store)
    _need(sorted(stores) == [0, 0, 36, 36], 'smart_fast_steps_every: stores per instance %r' % (stores,))


# ---- hand-over ------------------------------------------------------------------------------------------------------
def _is_wait_vm0(x):
    return x['op'] == 's_waitcnt' and 'vmcnt(0)' in x['args']


def _is_payload_access(x):
    """a vector-memory access that is neither a flag access (sc1) nor an atomic"""
    return x['cls'] == 'vmem' and x['op'].startswith(('global_load', 'global_store', 'flat_load', 'flat_store',
                                                      'buffer_load', 'buffer_store')) and 'sc1' not in x['args']


def lint_handover(lib=None, kernel=None):
    """  producer   s_waitcnt vmcnt(0) -> buffer_wbl2 sc1 -> s_waitcnt vmcnt(0) -> global_store_dword ... sc1 (the flag)
         consumer   global_load_dword ... sc1 (the poll) -> s_waitcnt vmcnt(0) -> buffer_inv sc1 -> plain loads
    at every publish and every wait of every sliced kernel; and no kernel that publishes is missing from SLICED."""
    dis = _disassembly(lib)
    for name in ([kernel] if kernel else SLICED):
        insts = dis.kernel(name)
        releases = [i for i, x in enumerate(insts) if x['op'] == 'buffer_wbl2']
        polls = [i for i, x in enumerate(insts) if x['op'] == 'global_load_dword' and 'sc1' in x['args']]
        # a sliced body publishes in two places (a slice that ran; a slice that gave up and poisons its chain) and waits
        # in one; the compiler may duplicate either, it may not lose one
        _need(len(releases) >= 2 and len(polls) >= 1, '%s: %d publishes, %d waits' % (name, len(releases), len(polls)))
        for i in releases:
            _need('sc1' in insts[i]['args'], '%s: buffer_wbl2 without sc1 (agent scope)' % name)
            # straight-line from the write-back to the flag store: a wait for it, and nothing that signals before the wait
            j = i + 1
            waited = False
            while j < len(insts) and not (insts[j]['op'].startswith(('global_store', 'global_atomic')) and
                                          ('sc1' in insts[j]['args'] or insts[j]['op'].startswith('global_atomic'))):
                waited = waited or _is_wait_vm0(insts[j])
                _need(waited or insts[j]['cls'] not in ('vmem', 'branch'),
                      '%s: %s %s between buffer_wbl2 and its wait' % (name, insts[j]['op'], insts[j]['args']))
                j += 1
            _need(j < len(insts) and waited, '%s: no s_waitcnt vmcnt(0) between buffer_wbl2 and the flag store' % name)
            _need(insts[j]['op'] == 'global_store_dword' and 'sc1' in insts[j]['args'], '%s: the flag store is %s' % (name, insts[j]['op']))
            # ... and ahead of the write-back the wave's own payload stores have been waited for
            k = i - 1
            while k >= 0 and insts[k]['cls'] not in ('vmem', 'branch'):
                if _is_wait_vm0(insts[k]):
                    break
                k -= 1
            _need(k >= 0 and _is_wait_vm0(insts[k]), '%s: payload stores not drained ahead of buffer_wbl2' % name)
        for i in polls:
            # behind the poll (in address order: the exit of the poll loop lies behind it) the L1 invalidate comes before
            # the first plain load of the hand-over; its own wait stands directly in front of it
            j = i + 1
            while j < len(insts) and insts[j]['op'] != 'buffer_inv':
                _need(not _is_payload_access(insts[j]),
                      '%s: %s %s between the poll and buffer_inv' % (name, insts[j]['op'], insts[j]['args']))
                j += 1
            _need(j < len(insts) and 'sc1' in insts[j]['args'], '%s: no buffer_inv sc1 behind the poll' % name)
            _need(_is_wait_vm0(insts[j - 1]), '%s: buffer_inv without its wait' % name)
    if kernel is None:
        names = fast_kernel_names()
        _need(len(names) >= 12 and set(SLICED) <= set(names), 'the list of sliced kernels does not match the library\'s')
        for name in names:
            publishes = any(x['op'] == 'buffer_wbl2' for x in dis.kernel(name))
            _need(publishes == (name in SLICED), '%s publishes but is not linted as a sliced kernel (or the reverse)' % name)


# ---- the row form's DPP chains ----------------------------------------------------------------------------------------
def _regs(token):
    """v[4:5] -> {4, 5}; v7 -> {7}; anything else -> empty (-v[2:3] and |v1| included)"""
    m = re.match(r'^[-|]*v\[(\d+):(\d+)\]\|?$', token)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'^[-|]*v(\d+)\|?$', token)
    return {int(m.group(1))} if m else set()


def _dest_regs(x):
    """the vector registers a vector instruction writes (its first operand), none for compares and stores"""
    if not x['op'].startswith('v_') or x['cls'] == 'vcmp' or x['op'].startswith(('v_readlane', 'v_readfirstlane')):
        return set()
    first = x['args'].split(',')[0].strip()
    return _regs(first)


def lint_rows(lib=None, kernel=None):
    """Every DPP instruction of the row-form kernels: no vector instruction within the two wait states in front of it
    writes the register it reads through DPP (src0; s_nop N counts N + 1), and no v_cmpx (a vector write of EXEC)
    within five.  Counted over the preceding instructions in address order AND, at the target of a branch, over the
    instructions in front of every branch that leads there.  Also: the kernels do hold the chains (>= 100 DPP
    instructions each), so that an empty check cannot pass."""
    dis = _disassembly(lib)
    for name in ([kernel] if kernel else ROWS):
        insts = dis.kernel(name)
        at = {x['addr']: i for i, x in enumerate(insts)}
        sources = {}
        for i, x in enumerate(insts):
            if x['cls'] == 'branch' and x['target'] in at:
                sources.setdefault(at[x['target']], []).append(i)

        def writers_before(i, need):
            """(instruction, wait states between it and instruction i) for every instruction that can execute within
            `need` wait states ahead of i, along fall-through and branches"""
            found, stack, seen = [], [(i, 0)], set()
            while stack:
                j, dist = stack.pop()
                for src in sources.get(j, []):          # arriving by a branch: the branch itself is one wait state
                    if (src, dist + 1) not in seen and dist + 1 <= need:
                        seen.add((src, dist + 1))
                        stack.append((src, dist + 1))
                k = j - 1
                if k < 0 or insts[k]['op'] in ('s_endpgm', 's_branch', 's_setpc_b64'):
                    continue                              # nothing falls through from there
                y = insts[k]
                if dist < need:
                    found.append((y, dist))
                step = 1 + (int(y['args'].strip() or 0) if y['op'] == 's_nop' else 0)
                if dist + step < need and (k, dist + step) not in seen:
                    seen.add((k, dist + step))
                    stack.append((k, dist + step))
            return found

        n_dpp = 0
        for i, x in enumerate(insts):
            if 'row_newbcast' not in x['args'] and 'row_shl' not in x['args'] and 'row_shr' not in x['args']:
                continue
            n_dpp += 1
            ops = [t.strip() for t in x['args'].split(' row_')[0].split(',')]
            src0 = _regs(ops[1]) if len(ops) > 1 else set()
            _need(src0, '%s: cannot read the DPP operand of %s %s' % (name, x['op'], x['args']))
            for y, dist in writers_before(i, 2):
                _need(not (_dest_regs(y) & src0),
                      '%s: %s %s at %#x reads through DPP what %s %s wrote %d wait state(s) earlier (two are needed)'
                      % (name, x['op'], x['args'], x['addr'], y['op'], y['args'], dist))
            for y, dist in writers_before(i, 5):
                _need(not y['op'].startswith('v_cmpx'),
                      '%s: %s at %#x follows a vector write of EXEC (%s) by %d wait states (five are needed)'
                      % (name, x['op'], x['addr'], y['op'], dist))
        _need(n_dpp >= 100, '%s holds %d DPP instructions: the row form is not in it' % (name, n_dpp))


# ---- the library as a whole -----------------------------------------------------------------------------------------
def sha256_of(path):
    h = hashlib.sha256()
    with open(path, 'rb') as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b''):
            h.update(chunk)
    return h.hexdigest()


def hipcc_version():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    try:
        out = subprocess.run([exe, '--version'], capture_output=True, text=True, timeout=60).stdout
        return ' | '.join(ln.strip() for ln in out.splitlines() if ln.strip())[:400]
    except Exception as e:      # noqa: BLE001
        return 'unknown (%s)' % e


def check_library(lib=LIB):
    """Run the three lints on `lib`.  -> {'sha256', 'hipcc', 'checked', 'pair_blocks', 'handover', 'rows', 'problems'}:
    each of the three True / False (None when llvm-objdump is missing: nothing checked)."""
    report = {'library': os.path.basename(lib), 'sha256': sha256_of(lib), 'hipcc': hipcc_version(), 'checked': False,
              'pair_blocks': None, 'handover': None, 'rows': None, 'problems': []}
    try:
        dis = Disassembly(lib)
    except LintError as e:
        report['problems'].append(str(e))
        return report
    report['checked'] = True
    for key, fn in (('pair_blocks', lint_pair_blocks), ('handover', lint_handover), ('rows', lint_rows)):
        try:
            fn(dis)
            report[key] = True
        except LintError as e:
            report[key] = False
            report['problems'].append('%s: %s' % (key, e))
    return report


def sidecar_path(lib=LIB):
    return lib + '.lint.json'


def write_sidecar(report, lib=LIB):
    tmp = '%s.%d.tmp' % (sidecar_path(lib), os.getpid())
    with open(tmp, 'w') as fh:
        json.dump(report, fh, indent=1)
        fh.write('\n')
    os.replace(tmp, sidecar_path(lib))


def read_sidecar(lib=LIB):
    try:
        with open(sidecar_path(lib)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def verdict_for(lib=LIB):
    """What smartpy_amd._lib does with the library it is about to load -> (ok_to_load, pair_blocks_ok, reason).
    The sidecar counts only for the very file it was written for (sha256)."""
    rep = read_sidecar(lib)
    if rep is None:
        return True, False, 'no lint record next to the library (%s): not built by smartpy_amd.build' % os.path.basename(sidecar_path(lib))
    if rep.get('sha256') != sha256_of(lib):
        return True, False, 'the lint record next to the library was written for another build of it'
    if not rep.get('checked'):
        return True, False, 'the library was built without llvm-objdump at hand: its code was not looked at'
    if rep.get('handover') is False or rep.get('rows') is False:
        return False, False, '; '.join(rep.get('problems', []))
    if not rep.get('pair_blocks'):
        return True, False, '; '.join(rep.get('problems', []))
    return True, True, ''


if __name__ == '__main__':
    import sys
    rep = check_library(sys.argv[1] if len(sys.argv) > 1 else LIB)
    print(json.dumps(rep, indent=1))
    sys.exit(0 if rep['checked'] and not rep['problems'] else 1)
