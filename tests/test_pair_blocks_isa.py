"""The pair blocks of the streaming step loop (smart_fast_arms.h: SMART_A_PAIRS_STRETCH), checked in the code the GPU
will run: smart_forcing_scan's code words are byte offsets into that code, computed from a stride and a block order
that the asm has to honour -- a block that outgrew its room or changed its place would send a jump into the middle of
another.  The built library is disassembled (hipcc cross-compiles here, no GPU needed) and every block looked at.
The checks themselves live in smartpy_amd/isa_lint.py: smartpy_amd.build runs them on every library it links (round 5),
these tests run them on the library of this tree -- and on a copy with one block offset corrupted."""
import os
import shutil

import pytest

from smartpy_amd import isa_lint

pytestmark = pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason='needs llvm-objdump of the ROCm toolchain')


@pytest.fixture(scope='module')
def dis():
    return isa_lint.Disassembly(isa_lint.LIB)


@pytest.mark.parametrize('kernel', isa_lint.PAIRED)
def test_every_pair_block_lies_where_the_code_words_point(dis, kernel):
    isa_lint.lint_pair_blocks(dis, kernel)


def test_every_block_of_the_every_step_stream_lies_where_its_code_word_points(dis):
    isa_lint.lint_every_stream(dis)


def test_a_dry_pair_is_eighteen_instructions_on_the_boundary(dis):
    """the shortest block, DD of the first position: 2 x 9 vector instructions, all 64-bit encodings on 8-byte addresses,
    then the two instructions of the jump"""
    stride = isa_lint._define('SMART_P_STRIDE')
    insts = dis.kernel('smart_fast_steps')
    at = {x['addr']: i for i, x in enumerate(insts)}
    i = [k for k in isa_lint._entries(insts) if not any(y['op'] == 's_load_dword' for y in insts[k:k + 12])][0]
    k = at[isa_lint._block_base(insts, i) + 4 * stride]
    body = insts[k:k + 20]
    assert [x['size'] for x in body[:18]] == [8] * 18 and all(x['addr'] % 8 == 0 for x in body[:18])
    assert all(x['op'].startswith(('v_mul_f64', 'v_fma_f64', 'v_add_f64')) for x in body[:18])
    assert [x['op'] for x in body[18:20]] == ['s_add_u32', 's_setpc_b64']


def _corrupt_one_block_offset(path):
    """Shift the base of the pair blocks of smart_fast_steps by 64 bytes: the literal of the `s_add_u32 s78, s78, <offset>`
    behind the loop's s_getpc_b64 (the encoding's second dword), found through the disassembly of the intact library."""
    dis = isa_lint.Disassembly(path)
    insts = dis.kernel('smart_fast_steps')
    i = [k for k in isa_lint._entries(insts) if not any(y['op'] == 's_load_dword' for y in insts[k:k + 12])][0]
    offset = int(insts[i + 1]['args'].split(',')[-1], 0)
    blob = open(path, 'rb').read()
    # s_add_u32 s78, s78, literal = SOP2 0x804eff4e followed by the 32-bit literal, little endian
    pattern = (0x804eff4e).to_bytes(4, 'little') + (offset & 0xffffffff).to_bytes(4, 'little')
    assert blob.count(pattern) >= 1
    at = blob.index(pattern)
    with open(path, 'wb') as fh:
        fh.write(blob[:at + 4] + ((offset + 64) & 0xffffffff).to_bytes(4, 'little') + blob[at + 8:])


def test_a_library_with_a_corrupted_block_offset_is_not_trusted_with_its_pair_blocks(tmp_path, monkeypatch):
    """What smartpy_amd.build does with every library it links, on a copy whose code words no longer fit its blocks: the
    lint names the kernel, the record written next to the library says the pair blocks are not to be used, and loading
    it switches the step loops to their threaded chunks (SMART_PAIR_BLOCKS=0) behind a warning -- never wrong code
    words.  The intact library passes and keeps its pair blocks."""
    from smartpy_amd import build, _lib
    good = str(tmp_path / 'libsmart_amd.so')
    shutil.copy(isa_lint.LIB, good)
    rep = build.lint(good)
    assert rep['checked'] and rep['pair_blocks'] and rep['handover'] and rep['rows'] and not rep['problems']
    isa_lint.write_sidecar(rep, good)
    assert isa_lint.verdict_for(good) == (True, True, '')
    bad = str(tmp_path / 'bad' / 'libsmart_amd.so')
    os.makedirs(os.path.dirname(bad))
    shutil.copy(isa_lint.LIB, bad)
    _corrupt_one_block_offset(bad)
    with pytest.raises(isa_lint.LintError, match='smart_fast_steps'):
        isa_lint.lint_pair_blocks(bad)
    rep = build.lint(bad)                   # (pair blocks that fail do not refuse the library: they are recorded)
    assert rep['pair_blocks'] is False and rep['handover'] and rep['rows'] and 'smart_fast_steps' in rep['problems'][0]
    isa_lint.write_sidecar(rep, bad)
    ok, pairs, reason = isa_lint.verdict_for(bad)
    assert ok and not pairs and 'smart_fast_steps' in reason
    # the record of ANOTHER file does not count for this one, and no record at all means no pair blocks either
    shutil.copy(isa_lint.sidecar_path(good), isa_lint.sidecar_path(bad))
    assert isa_lint.verdict_for(bad)[1] is False and 'another build' in isa_lint.verdict_for(bad)[2]
    os.remove(isa_lint.sidecar_path(bad))
    assert isa_lint.verdict_for(bad)[1] is False
    # ... which is what the loader acts on
    monkeypatch.setattr(_lib, 'LIB_PATH', bad)
    monkeypatch.delenv('SMART_PAIR_BLOCKS', raising=False)
    with pytest.warns(UserWarning, match='threaded chunks'):
        _lib._apply_lint_verdict()
    assert os.environ['SMART_PAIR_BLOCKS'] == '0'
    monkeypatch.setenv('SMART_PAIR_BLOCKS', '1')        # the caller's own setting stands (A/B builds)
    _lib._apply_lint_verdict()
    assert os.environ['SMART_PAIR_BLOCKS'] == '1'
    monkeypatch.delenv('SMART_PAIR_BLOCKS', raising=False)
    monkeypatch.setattr(_lib, 'LIB_PATH', good)
    _lib._apply_lint_verdict()
    assert 'SMART_PAIR_BLOCKS' not in os.environ


def test_a_library_whose_hand_over_fails_the_lint_is_refused(tmp_path, monkeypatch):
    from smartpy_amd import build, _lib
    lib = str(tmp_path / 'libsmart_amd.so')
    shutil.copy(isa_lint.LIB, lib)
    monkeypatch.setattr(isa_lint, 'lint_handover',
                        lambda dis: (_ for _ in ()).throw(isa_lint.LintError('smart_fast_steps: no s_waitcnt vmcnt(0)')))
    with pytest.raises(build.BuildLintError, match='not installed'):
        build.lint(lib)
    rep = isa_lint.check_library(lib)
    isa_lint.write_sidecar(rep, lib)
    monkeypatch.setattr(_lib, 'LIB_PATH', lib)
    with pytest.raises(ImportError, match='failed the code lints'):
        _lib._apply_lint_verdict()


def test_the_lint_verdict_travels_inside_the_library(tmp_path, monkeypatch):
    """Round 6: a caller of the C ABI never reads the record next to the library, so the verdict on the pair blocks is
    written INTO the file (smart_capi.hip: smart_lint_stamp; smartpy_amd.build stamps what it has linted and found in
    order).  The library of this tree says so in smart_build_info(); a copy whose stamp is back to 'unchecked' -- what a
    build by another route carries -- runs its threaded chunks by itself, and the loader, once the lint has looked at
    that copy's code, gives this process its word for the pair blocks (SMART_PAIR_BLOCKS=1)."""
    import ctypes
    from smartpy_amd import _lib
    assert isa_lint.library_stamp(isa_lint.LIB) == 'pairs-ok'
    info = ctypes.CDLL(isa_lint.LIB).smart_build_info
    info.restype = ctypes.c_char_p
    assert info().decode().endswith('SMART_LINT_STAMP=pairs-ok') and 'ABI 7' in info().decode()
    # a library that was never stamped (another build route)
    plain = str(tmp_path / 'libsmart_amd.so')
    with open(isa_lint.LIB, 'rb') as fh:
        blob = fh.read()
    assert blob.count(isa_lint.STAMP_PAIRS_OK) == 1
    with open(plain, 'wb') as fh:
        fh.write(blob.replace(isa_lint.STAMP_PAIRS_OK, isa_lint.STAMP_UNCHECKED))
    assert isa_lint.library_stamp(plain) == 'unchecked'
    # (in a process of its own: a second copy of the library in this one would register its code objects twice)
    import subprocess
    import sys
    out = subprocess.run([sys.executable, '-c', 'import ctypes, sys; f = ctypes.CDLL(sys.argv[1]).smart_build_info; '
                          'f.restype = ctypes.c_char_p; print(f().decode())', plain], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith('SMART_LINT_STAMP=unchecked'), out.stdout + out.stderr
    # stamping: only a passed lint writes 'pairs-ok'; the stamp is found exactly once
    assert isa_lint.stamp_library(plain, False) and isa_lint.library_stamp(plain) == 'unchecked'
    monkeypatch.setattr(_lib, 'LIB_PATH', plain)
    monkeypatch.delenv('SMART_PAIR_BLOCKS', raising=False)
    _lib._apply_lint_verdict()              # no record: looks at the code (under the lock), finds the blocks in order
    assert os.environ['SMART_PAIR_BLOCKS'] == '1' and isa_lint.read_sidecar(plain)['pair_blocks'] is True
    monkeypatch.delenv('SMART_PAIR_BLOCKS', raising=False)
    assert isa_lint.stamp_library(plain, True) and isa_lint.library_stamp(plain) == 'pairs-ok'
    assert not isa_lint.stamp_library(plain, True)          # (nothing left to stamp)
