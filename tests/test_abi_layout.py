"""CPU: the ctypes mirrors of bmt_amd/_lib.py have the layout the C compiler gives the structs of include/bmt_hip.h.

A field added to one side only does not fail at load time -- every later field is read at the wrong offset and the launch computes garbage (or
faults) on the GPU box.  Here the header is compiled by gcc into a program that prints sizeof / offsetof of every mirrored struct and field."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MIRRORS = {          # C struct -> ctypes class
    "bmt_gemm_bf16_args": "GemmBf16Args",
    "bmt_gemm_batch": "GemmBatch",
    "bmt_attn_fwd_args": "AttnFwdArgs",
    "bmt_attn_bwd_args": "AttnBwdArgs",
    "bmt_attn_fwd_bf16_args": "AttnFwdBf16Args",
    "bmt_attn_bwd_bf16_args": "AttnBwdBf16Args",
    "bmt_colsum_item": "ColsumItem",
    "bmt_copy_item": "CopyItem",
    "bmt_select_proposals_args": "SelectProposalsArgs",
}


def test_struct_layouts_match_the_header(tmp_path):
    from bmt_amd import _lib
    header = open(os.path.join(ROOT, "include", "bmt_hip.h")).read()
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "bmt_hip.h"', "int main(void) {"]
    want = {}
    for cname, pyname in MIRRORS.items():
        if ("} " + cname + ";") not in header:
            pytest.fail(f"{cname} is not a struct of include/bmt_hip.h any more: update MIRRORS")
        cls = getattr(_lib, pyname)
        lines.append(f'    printf("{cname} %zu\\n", sizeof({cname}));')
        want[cname] = C.sizeof(cls)
        for fname, _ in cls._fields_:
            lines.append(f'    printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
            want[f"{cname}.{fname}"] = getattr(cls, fname).offset
    lines += ["    return 0;", "}"]
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text("\n".join(lines))
    r = subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, "a ctypes mirror names a field the header's struct does not have (or the header is not plain C):\n" + r.stderr[-2000:]
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    bad = {k: (int(got[k]), v) for k, v in want.items() if int(got[k]) != v}
    assert not bad, f"(C, ctypes) disagree: {bad}"


def test_every_argument_struct_of_the_header_is_mirrored():
    """a struct the header gains must get a mirror (and a row above), or it is not callable from the host side at all"""
    import re
    header = open(os.path.join(ROOT, "include", "bmt_hip.h")).read()
    structs = set(re.findall(r"^\}\s*(bmt_[a-z0-9_]+);", header, flags=re.M))
    known = set(MIRRORS)
    assert structs <= known, f"structs of the header without a layout check: {sorted(structs - known)}"
