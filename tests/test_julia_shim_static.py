"""Static check of the Julia shim (which no test can execute: there is no Julia here): every `ccall` names a symbol
that include/tempest_hip.h declares, passes as many arguments as the prototype takes, and uses a Julia C type of the
same class (pointer / integer width / float width) as the C parameter at every position; the return type agrees too."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_class(t):
    t = re.sub(r"\bconst\b", "", t).strip()
    if "*" in t:
        return "ptr"
    t = t.split()[:-1] if len(t.split()) > 1 else t.split()  # drop the parameter name
    t = " ".join(t) if t else ""
    return {"int": "i32", "unsigned": "i32", "size_t": "i64", "float": "f32", "double": "f64", "void": "void",
            "unsigned long long": "i64"}.get(t, t)


def jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring",):
        return "ptr"
    return {"Cint": "i32", "Cuint": "i32", "Csize_t": "i64", "Cfloat": "f32", "Float32": "f32", "Cdouble": "f64", "Float64": "f64",
            "Cvoid": "void", "Culonglong": "i64", "UInt64": "i64", "Int32": "i32"}.get(t, t)


def header_protos():
    src = open(os.path.join(ROOT, "include", "tempest_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(tsdr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else split_args(args)
        rc = "ptr" if "*" in ret else c_class(ret + " x")
        protos[name] = (rc, [c_class(p) for p in params])
    return protos


def shim_ccalls():
    src = open(os.path.join(ROOT, "tempestsdr.jl_amd", "julia", "TempestHIP.jl")).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(tsdr_[a-z0-9_]+),\s*LIB\),", src):
        # the balanced argument list of this ccall
        i = src.index("(", m.start())
        depth, j = 0, i
        while True:
            depth += src[j] == "("
            depth -= src[j] == ")"
            if depth == 0:
                break
            j += 1
        parts = split_args(src[i + 1:j])
        ret, types = parts[1], parts[2]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tl = [t for t in split_args(types[1:-1]) if t]
        calls.append((m.group(1), jl_class(ret), [jl_class(t) for t in tl], len(parts) - 3, src.count("\n", 0, m.start()) + 1))
    return calls


def test_every_ccall_matches_its_prototype():
    protos = header_protos()
    calls = shim_ccalls()
    assert len(calls) >= 25
    for name, ret, types, nargs, line in calls:
        assert name in protos, f"TempestHIP.jl:{line}: {name} is not declared in include/tempest_hip.h"
        cret, cparams = protos[name]
        assert len(types) == len(cparams), f"TempestHIP.jl:{line}: {name} takes {len(cparams)} arguments, the ccall lists {len(types)} types"
        assert nargs == len(types), f"TempestHIP.jl:{line}: {name}: {len(types)} types but {nargs} values"
        assert ret == cret, f"TempestHIP.jl:{line}: {name} returns {cret}, the ccall says {ret}"
        for k, (a, b) in enumerate(zip(types, cparams)):
            assert a == b, f"TempestHIP.jl:{line}: {name} argument {k + 1}: C {b}, Julia {a}"
