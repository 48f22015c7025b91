"""The C ABI is the drop-in boundary: every function declared in include/capmi.h must be exported by
the built library and bound (with a matching argument count) in imagecaptioning/pytorch_amd/_lib.py.
CPU only: the library is loaded, no compute entry point is called."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'capmi.h')
LIB = os.path.join(ROOT, 'imagecaptioning', 'pytorch_amd', 'libcapmi.so')


def declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|int64_t|const char \*)\s*(capmi_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',')])
        out[name] = n
    return out


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(LIB):
        from imagecaptioning.pytorch_amd import build
        build.build(verbose=False)
    return LIB


def test_header_declares_the_hot_path_entry_points():
    d = declared()
    for name in ('capmi_gemm_f32', 'capmi_attention_fwd', 'capmi_attention_bwd', 'capmi_lstm_cell_fwd', 'capmi_lstm_cell_bwd',
                 'capmi_logsoftmax_select', 'capmi_ciderd_score', 'capmi_updown_rollout_fwd', 'capmi_updown_rollout_bwd',
                 'capmi_adam_step'):
        assert name in d, name


def test_library_exports_every_declared_symbol(built):
    out = subprocess.check_output(['nm', '-D', '--defined-only', built]).decode()
    exported = set(re.findall(r' T (capmi_\w+)', out))
    missing = set(declared()) - exported
    assert not missing, 'declared in capmi.h but not exported: %s' % sorted(missing)


def test_ctypes_table_matches_header(built):
    from imagecaptioning.pytorch_amd import _lib
    d = declared()
    assert set(_lib.SIGNATURES) == set(d), (set(_lib.SIGNATURES) ^ set(d))
    for name, n in d.items():
        assert len(_lib.SIGNATURES[name]) == n, '%s: header has %d args, binding %d' % (name, n, len(_lib.SIGNATURES[name]))
    assert _lib.lib.capmi_arch() == b'gfx950'
    assert _lib.lib.capmi_version() >= 1


def test_struct_layouts_match_header():
    """Field order of the ctypes structs == field order in the header (they are filled by name in Python
    and read by offset in C++)."""
    from imagecaptioning.pytorch_amd import _lib
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)

    def fields(struct):
        body = re.search(r'typedef struct (?:%s )?\{([^{}]*?)\} %s;' % (struct, struct), src, flags=re.S).group(1)
        names = []
        for stmt in body.split(';'):
            stmt = stmt.strip()
            if not stmt:
                continue
            for part in stmt.split(','):
                nm = re.findall(r'(\w+)\s*(?:\[\w+\])?$', part.strip())[0]
                names.append(nm)
        return names

    pairs = {'capmi_gemm_seg': _lib.GemmSeg, 'capmi_gemm_desc': _lib.GemmDesc, 'capmi_updown_weights': _lib.UpDownWeights,
             'capmi_updown_rollout': _lib.UpDownRollout, 'capmi_updown_grads': _lib.UpDownGrads,
             'capmi_updown_bwd_scratch': _lib.UpDownBwdScratch, 'capmi_sparse_logp_grad': _lib.SparseLogpGrad, 'capmi_mask_desc': _lib.MaskDesc,
             'capmi_reduce_item': _lib.ReduceItem, 'capmi_colsum_item': _lib.ColsumItem,
             'capmi_newfc_bwd_scratch': _lib.NewFCBwdScratch, 'capmi_next_embed': _lib.NextEmbed, 'capmi_sample_filter': _lib.SampleFilter,
             'capmi_step_state': _lib.StepState, 'capmi_group_gemm': _lib.GroupGemm}
    for cname, cls in pairs.items():
        assert fields(cname) == [f[0] for f in cls._fields_], cname


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure; the product package must not reference it."""
    pkg = os.path.join(ROOT, 'imagecaptioning')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.h'):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), os.path.join(dirpath, f)
