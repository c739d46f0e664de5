"""The C-ABI library loads and exports every symbol include/rpo_hip.h declares (no GPU needed, no compute calls)."""
import ctypes
import os
import subprocess

import pytest

from rpo_amd import _lib


def test_header_parses():
    protos, consts = _lib.parse_header()
    assert len(protos) >= 20
    for must in ("rpo_cartsafe_step", "rpo_cartsafe_act_project", "rpo_pendulum_step", "rpo_replay_sample_gather",
                 "rpo_td_huber", "rpo_adam_step", "rpo_cartsafe_lagrangian", "rpo_abi_version"):
        assert must in protos
    assert consts["RPO_CART_ROW"] == 24 and consts["RPO_PEND_ROW"] == 16 and consts["RPO_CART_RING"] == 32 and consts["RPO_PEND_RING"] == 16 and consts["RPO_ERR_ARG"] == -1
    # spot-check the type mapping
    assert protos["rpo_polyak"] == [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIBRARY):
        from rpo_amd.csrc import build
        build.build(verbose=False)
    lib = _lib.load()
    for name in _lib.PROTOTYPES:
        assert hasattr(lib, name), name
    assert lib.rpo_abi_version() == _lib.CONST["RPO_ABI_VERSION"]
    # nothing torch-typed in the dynamic symbol table: the boundary is plain C
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIBRARY], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in syms.splitlines() if " T " in line}
    assert set(_lib.PROTOTYPES) <= exported
    # ... and nothing else: the dynamic symbol table IS the header (internal cross-unit functions are visibility("hidden"))
    assert exported == set(_lib.PROTOTYPES), sorted(exported - set(_lib.PROTOTYPES))


def test_argument_validation_without_gpu():
    """Argument checks run before any HIP call, so they can be exercised on a CPU-only box."""
    lib = _lib.load()
    assert lib.rpo_polyak(0, None, None, 0.5, None) == _lib.CONST["RPO_ERR_ARG"]
    assert lib.rpo_polyak(4, None, None, 0.5, None) == _lib.CONST["RPO_ERR_NULL"]
    assert lib.rpo_replay_gather(None, 32, 23, 4, None, None, None) == _lib.CONST["RPO_ERR_ARG"]   # row not 16-B multiple
    assert lib.rpo_replay_gather(None, 16, 24, 4, None, None, None) == _lib.CONST["RPO_ERR_ARG"]   # ring stride < row
    with pytest.raises(_lib.RpoHipError):
        _lib.check(-1, "x")


def test_ops_refuse_cpu_tensors():
    import torch
    from rpo_amd import ops
    with pytest.raises(_lib.RpoHipError):
        ops.polyak(torch.zeros(4), torch.zeros(4), 0.5)


def test_struct_layouts_match_the_compiler(tmp_path):
    """The ctypes mirrors of the header's structs (rpo_split_update, rpo_adam_seg, rpo_mlp, rpo_td) have the compiler's
    size and field offsets: a C program that includes include/rpo_hip.h prints them, gcc only."""
    import torch  # noqa: F401  (rpo_amd.ops imports it)
    from rpo_amd import ops
    structs = {"rpo_split_update": ops._SplitUpdateStruct, "rpo_adam_seg": ops._AdamSegStruct, "rpo_mlp": ops._MlpStruct,
               "rpo_mlp_grad": ops._MlpGradStruct, "rpo_td": ops._TdStruct, "rpo_rollout_rider": ops._RolloutRiderStruct}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % _lib.HEADER, "int main(void) {"]
    for cname, st in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, field, value = line.split()
        st = structs[cname]
        if field == "size":
            assert ctypes.sizeof(st) == int(value), (cname, ctypes.sizeof(st), value)
        else:
            assert getattr(st, field).offset == int(value), (cname, field)
        seen += 1
    assert seen > 120


def test_split_update_stages_validate_arguments():
    lib = _lib.load()
    for stage in ("critic_fwd_a", "critic_fwd_b", "critic_fwd_b_pol", "pend_head_project", "critic_bwd_a", "critic_bwd_b", "policy_a",
                  "policy_b", "policy_c", "policy_d", "policy_e", "critic_front", "critic_front_pol", "critic_mid", "critic_mid_pol",
                  "critic_pfront", "critic_pfront_pol", "policy_front", "policy_front_bc"):
        assert getattr(lib, "rpo_split_" + stage)(None, None) == _lib.CONST["RPO_ERR_NULL"]
    assert lib.rpo_xcc_probe(8, 16, 4, 256, None, None) == _lib.CONST["RPO_ERR_NULL"]
    assert lib.rpo_xcc_probe(0, 16, 4, 256, None, None) == _lib.CONST["RPO_ERR_ARG"]
    for stage in ("critic_fwd_a_ride", "critic_fwd_b_ride", "critic_bwd_b_ride", "critic_front_ride", "critic_mid_ride", "critic_pfront_ride"):            # the riding rollout halves: both structs are required
        assert getattr(lib, "rpo_split_" + stage)(None, None, None) == _lib.CONST["RPO_ERR_NULL"]
    assert _lib.CONST["RPO_ABI_VERSION"] == 6


def test_every_entry_point_survives_null_arguments():
    """Every `int rpo_*(...)` of the header called (a) with all-zero arguments and (b) with sizes = 4, scalars = 1 and NULL
    pointers: the validation in front of every launch must answer RPO_ERR_ARG / RPO_ERR_NULL (or, at worst, a HIP error code
    from a launch attempt on a box without a GPU) -- never dereference a NULL host pointer.  Run under ASan + UBSan by
    tests/test_sanitizers.py."""
    lib = _lib.load()
    skip = {"rpo_abi_version", "rpo_tuning", "rpo_mlp_supported", "rpo_mlp_split_supported"}    # (predicates: 0 = "no")
    assert lib.rpo_mlp_supported(0, 0, 0) == 0 and lib.rpo_mlp_split_supported(None) == 0
    bad = {}
    for name, argtypes in _lib.PROTOTYPES.items():
        if name in skip:
            continue
        fn = getattr(lib, name)
        for fill in (0, 4):
            args = []
            for t in argtypes:
                if t is ctypes.c_void_p:
                    args.append(None)
                elif t is ctypes.c_float:
                    args.append(float(fill and 1.0))
                else:
                    args.append(fill)
            rc = fn(*args)
            if rc == 0:
                bad[(name, fill)] = rc
    assert not bad, "entry points that report success on NULL / empty arguments: %s" % bad
    # the switch table: unknown keys are refused, a query does not change the value
    assert lib.rpo_tuning(-1, 1) == _lib.CONST["RPO_ERR_ARG"] and lib.rpo_tuning(_lib.CONST["RPO_TUNE_COUNT"], 1) == _lib.CONST["RPO_ERR_ARG"]
    was = lib.rpo_tuning(_lib.CONST["RPO_TUNE_BWD_ONEPASS"], -1)
    assert lib.rpo_tuning(_lib.CONST["RPO_TUNE_BWD_ONEPASS"], 0) == was and lib.rpo_tuning(_lib.CONST["RPO_TUNE_BWD_ONEPASS"], was) == 0
    assert lib.rpo_tuning(_lib.CONST["RPO_TUNE_BWD_ONEPASS"], -1) == was
