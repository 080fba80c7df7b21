"""The C-ABI library loads and exports every symbol include/gcnvae.h declares; the product never
imports the oracle.  No compute calls (CPU box)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'gcnvae.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gv_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from gcn_vae_amd import lib
    names = declared_symbols()
    assert len(names) >= 28
    handle = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f'{n} declared in include/gcnvae.h but not exported'
    assert sorted(lib.SIGNATURES) == names, 'ctypes signature table out of sync with the header'
    assert lib.load().gv_version() >= 100
    assert lib.last_error() == '' or isinstance(lib.last_error(), str)


def test_argument_errors_are_reported_not_thrown():
    from gcn_vae_amd import lib
    l = lib.load()
    rc = l.gv_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, 1, None, None, 0, None)
    assert rc == -1 and 'NULL' in lib.last_error()
    rc = l.gv_segment_items_count(None, 3, 0, None, None, None, None)
    assert rc < 0
    # the MADE entry points of round 2: descriptor tables are validated on the host, before anything is launched
    from gcn_vae_amd import ops
    layers = (ops._ChainLayer * 2)()
    assert l.gv_made_chain(None, 200, 64, 2, ctypes.addressof(layers), None) != 0 and 'NULL' in lib.last_error()
    assert l.gv_made_chain(None, 200, 64, 9, ctypes.addressof(layers), None) != 0            # more layers than the table holds
    assert l.gv_made_chain(None, 200, 0, 2, ctypes.addressof(layers), None) == 0             # no rows: nothing to do
    n_ok = (ctypes.c_int32 * 2)(200, 400)
    k_ok = (ctypes.c_int32 * 2)(200, 200)
    k_bad = (ctypes.c_int32 * 2)(200, 208)
    wide = (ctypes.c_int32 * 2)(2048, 2048)
    assert l.gv_made_chain_fits(2, ctypes.addressof(n_ok), ctypes.addressof(k_ok), 1) == 1
    assert l.gv_made_chain_fits(2, ctypes.addressof(n_ok), ctypes.addressof(k_bad), 0) == 0   # k of a layer != n of the one before
    assert l.gv_made_chain_fits(2, ctypes.addressof(wide), ctypes.addressof(wide), 0) == 0    # tiles larger than the LDS
    assert l.gv_made_pack_weight_elems(200, 200) == 7 * 13 * 64 * 8
    rows = (ops._RowLayer * 1)()
    assert l.gv_made_row_fwd(None, 1, ctypes.addressof(rows), None) != 0
    assert l.gv_mul_multi(9, None, None, None, None, None) != 0 and l.gv_mul_multi(0, None, None, None, None, None) == 0
    outs = (ctypes.c_void_p * 2)()
    segs = (ctypes.c_int32 * 2)(3, 4)
    assert l.gv_rowsum_bf16_segments(None, 8, 7, 8, 2, ctypes.addressof(outs), ctypes.addressof(segs), 1, None, None) != 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gcn-vae_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle/' not in src or f.endswith('.md'), f


def test_header_is_plain_c_and_struct_layouts_match_the_ctypes_mirrors(tmp_path):
    """include/gcnvae.h compiles as C99 on its own (no C++, no HIP, no torch types), and the two descriptor structs the MADE
    entry points take (gv_chain_layer, gv_row_layer) and the batched index builder's gv_csr_job have the size and field offsets of their ctypes mirrors in ops.py."""
    import shutil
    import subprocess
    import pytest
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    from gcn_vae_amd import ops
    fields = {'gv_chain_layer': ops._ChainLayer, 'gv_row_layer': ops._RowLayer, 'gv_csr_job': ops.batch_index._CsrJob}
    lines = []
    for name, cls in fields.items():
        lines.append(f'printf("{name} %zu", sizeof({name}));')
        for f, _ in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({name}, {f}));')
        lines.append('printf("\\n");')
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gcnvae.h"\nint main(void) {\n' + '\n'.join(lines) + '\nreturn 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        name, size, *offs = line.split()
        cls = fields[name]
        assert int(size) == ctypes.sizeof(cls), name
        assert [int(o) for o in offs] == [getattr(cls, f).offset for f, _ in cls._fields_], name


def test_work_item_size_follows_the_edge_list(monkeypatch):
    """ops.chunk_for: items of 16 edges for a sampled batch, 128 at FB15k-237 size, 256 from a million entries on, powers of two between;
    the by-relation lists are capped at their own default; GV_CHUNK overrides."""
    from gcn_vae_amd import ops
    monkeypatch.delenv('GV_CHUNK', raising=False)
    assert ops.chunk_for(0) == 16 and ops.chunk_for(20000) == 16 and ops.chunk_for(131072) == 32
    assert ops.chunk_for(173670) == 32 and ops.chunk_for(440000) == 64
    assert ops.chunk_for(544230) == 128 and ops.chunk_for(1_048_576) == 256 and ops.chunk_for(50_000_000) == 256
    assert ops.chunk_for(50_000_000, ops.DEFAULT_CHUNK_REL) == 128 and ops.chunk_for(20000, ops.DEFAULT_CHUNK_REL) == 16
    assert ops.chunk_for(173670, ops.DEFAULT_CHUNK_REL) == 128 and ops.chunk_for(544230, ops.DEFAULT_CHUNK_REL) == 128
    cs = [ops.chunk_for(n) for n in range(0, 2_000_000, 4099)]
    assert all(c & (c - 1) == 0 and 16 <= c <= 256 for c in cs) and cs == sorted(cs)
    monkeypatch.setenv('GV_CHUNK', '48')
    assert ops.chunk_for(10) == 48 and ops.chunk_for(10_000_000) == 48


def test_round3_entry_points_validate_their_arguments_on_the_host():
    """The entry points added in round 3 report bad arguments through the return code (nothing is launched: runs without a GPU)."""
    from gcn_vae_amd import lib, ops
    l = lib.load()
    assert l.gv_gather3_i32(None, 5, None, None, None, None, None, None, None) != 0 and 'NULL' in lib.last_error()
    assert l.gv_gather3_i32(None, 0, None, None, None, None, None, None, None) == 0                      # nothing to do
    assert l.gv_iaf_update_bwd_row0(None, None, None, None, None, None, None, None, 10, 202, None) != 0    # d % 4 != 0
    assert l.gv_iaf_update_bwd_row0(None, None, None, None, None, None, None, None, 10, 200, None) != 0 and 'NULL' in lib.last_error()
    assert l.gv_iaf_update_bwd_row0_workspace_floats(200) == 1024 * 400
    assert l.gv_iaf_update_bwd_bf16_ex(None, None, 200, None, None, None, None, None, 400, None, 16, None, 0, 0, 200, None) == 0   # n == 0
    assert l.gv_iaf_update_bwd_bf16_ex(None, None, 200, None, None, None, None, None, 400, None, 16, None, 0, 8, 200, None) != 0
    assert l.gv_made_pack_weight_iaf(None, 200, 400, 200, None, None) != 0
    assert l.gv_made_pack_weight_iaf(None, 200, 404, 200, None, None) != 0 and 'n=' in lib.last_error()    # n = 2 d, d % 8 == 0
    plan = (ctypes.c_int32 * 3)()
    assert l.gv_rgcn_bdd_lds_plan(20, 10, 10, 22, 0, ctypes.addressof(plan)) in (0, 1)
    # a chain layer that asks for mask bits together with a tile mask is refused before any launch
    layers = (ops._ChainLayer * 1)()
    layers[0].n, layers[0].k = 200, 200
    assert l.gv_made_chain(None, 200, 64, 1, ctypes.addressof(layers), None) != 0
    # the 64-row-tile forms: tile distances are checked against the row counts before anything is launched
    assert l.gv_gemm_bf16_gradw_tiles(None, 64 * 200, None, 64 * 200, 200, 200, 4096, None, 1, None, 8, None, 0, None) != 0 and 'NULL' in lib.last_error()
    assert l.gv_iaf_update_fwd_bf16_tiles(None, None, 400, None, None, None, None, 200, None, 64 * 200, 0, 200, None) == 0    # n == 0
    assert l.gv_iaf_update_fwd_bf16_tiles(None, None, 400, None, None, None, None, 200, None, 0, 10, 200, None) != 0 and 't_tile' in lib.last_error()
