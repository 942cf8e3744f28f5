"""Deck function strings (MHA_FUNC_EXPRESSION): the oracle's recursive-descent evaluator against Python on the strings of
the mirrored reference decks, grammar corner cases, and the product's host compiler accepting / rejecting the same
vocabulary (no GPU needed)."""
import math
import os
import re

import numpy as np
import pytest
import yaml

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")


def deck_strings():
    """Every function-like string of the mirrored input decks (Functions:, Neumann / Dirichlet data, true solutions)."""
    out = set()

    def walk(node):
        if isinstance(node, dict):
            for v in node.values():
                walk(v)
        elif isinstance(node, str) and re.search(r"[xyzt]|pi|\d", node) and re.fullmatch(r"[-+*/^().\w\s<>=]+", node):
            out.add(node)

    for f in sorted(os.listdir(GOLD)):
        if f.endswith(".input.yaml"):
            d = yaml.safe_load(open(os.path.join(GOLD, f)))["ANONYMOUS"]
            walk(d.get("Functions", {}))
            walk(d.get("Physics", {}))
            walk(d.get("Postprocess", {}).get("True solutions", {}))
    return sorted(s for s in out if not re.fullmatch(r"[A-Za-z ]+", s))


def py_eval(s, x, y, z, t, nx=0.0, ny=0.0):
    env = {k: getattr(math, k) for k in ("sin", "cos", "tan", "exp", "log", "sqrt", "sinh", "cosh")}
    env.update(abs=abs, pi=math.pi, x=x, y=y, z=z, t=t, nx=nx, ny=ny, nz=0.0, h=0.0)
    return float(eval(s.replace("^", "**"), {"__builtins__": {}}, env))


def test_deck_strings_evaluate(oracle):
    import mrhyde_amd
    strings = [s for s in deck_strings() if not re.search(r"\b(thermal|navier|porous|Stokes|true|false|quad|hex)\b", s)]
    assert any("sin(2*pi*x)" in s for s in strings) and any("cos(2*pi*y)" in s for s in strings)
    rng = np.random.default_rng(1)
    for s in strings:
        try:
            py_eval(s, 0.3, 0.4, 0.5, 0.1)
        except Exception:
            continue  # not a function string (module names, solver options)
        mrhyde_amd.api.check_expression(s)  # the product's compiler accepts every deck function
        for _ in range(5):
            x, y, z, t = rng.uniform(0, 1, 4)
            ref = py_eval(s, x, y, z, t, 0.6, -0.8)
            got = oracle.eval_expression(s, [x, y, z], t, nrm=[0.6, -0.8])
            assert abs(got - ref) <= 1e-13 * max(1.0, abs(ref)), s


@pytest.mark.parametrize("s,val", [("-2^2", -4.0), ("2^-1", 0.5), ("2^3^2", 512.0), ("-(1+2)*3", -9.0), ("1-2-3", -4.0),
                                   ("8/4/2", 1.0), ("(x<0.5)*3+(x>=0.5)*7", 7.0), ("1.5e1+.5", 15.5), ("abs(-x)+sqrt(4)", 2.75),
                                   ("exp(log(3))", 3.0), ("sinh(0)+cosh(0)", 1.0), ("2*pi", 2 * math.pi), ("+x", 0.75)])
def test_grammar_corner_cases(oracle, s, val):
    import mrhyde_amd
    assert abs(oracle.eval_expression(s, [0.75, 0.0, 0.0]) - val) < 1e-13
    mrhyde_amd.api.check_expression(s)


@pytest.mark.parametrize("s", ["2*e", "max(x)", "sin x", "(1+2", "1+", "3 4", "x**2", "emean(x)", ""])
def test_unsupported_strings_are_rejected(oracle, s):
    import mrhyde_amd
    with pytest.raises(mrhyde_amd.MhaError) as e:
        mrhyde_amd.api.check_expression(s)
    assert e.value.code == 1
    if s not in ("x**2",):
        with pytest.raises(ValueError):
            oracle.eval_expression(s, [0.1, 0.2, 0.3])


def test_field_dependent_deck_strings_restatement(oracle):
    """The numpy restatement of FunctionManager<AD>::evaluate (oracle_lib.deck_eval_ad / assemble_thermal_fields) pinned on
    the CPU: with constant strings it reproduces the C oracle's thermal assembly (itself pinned by the reference's golds);
    with 'thermal diffusion' = "1+e*e" the Jacobian is the central difference of the residual, and a named function
    referenced from another string ("kappa0 + e*e") gives the same result as the inlined string."""
    dim, order, qdeg = 2, 2, 4
    m = oracle.mesh_structured(dim, order, (3, 2))
    rng = np.random.default_rng(12)
    nodes = m["nodes"] + 0.03 * rng.uniform(-1, 1, m["nodes"].shape)
    u = rng.uniform(-1, 1, m["ndof"])
    ref = oracle.assemble_thermal(dim, order, qdeg, nodes, m["lids"], m["offsets"], u, diff=1.3, source=("const", 0.7))
    got = oracle.assemble_thermal_fields(dim, order, qdeg, nodes, m["lids"], m["offsets"], u,
                                         {"thermal source": "0.7", "thermal diffusion": "1.3"}, rowptr=ref["rowptr"], colind=ref["colind"])
    assert np.abs(got["crs_vals"] - ref["crs_vals"]).max() < 1e-13 * np.abs(ref["crs_vals"]).max()
    assert np.abs(got["res"] - ref["res"]).max() < 1e-13 * np.abs(ref["res"]).max()
    funcs = {"thermal source": "2*sin(pi*x)*y + 0.1*e", "thermal diffusion": "1+e*e + 0.2*grad(e)[x]^2"}
    out = oracle.assemble_thermal_fields(dim, order, qdeg, nodes, m["lids"], m["offsets"], u, funcs, rowptr=ref["rowptr"], colind=ref["colind"])
    v = rng.uniform(-1, 1, m["ndof"])
    eps = 1e-6
    rp = oracle.assemble_thermal_fields(dim, order, qdeg, nodes, m["lids"], m["offsets"], u + eps * v, funcs, rowptr=ref["rowptr"], colind=ref["colind"])["res"]
    rm = oracle.assemble_thermal_fields(dim, order, qdeg, nodes, m["lids"], m["offsets"], u - eps * v, funcs, rowptr=ref["rowptr"], colind=ref["colind"])["res"]
    import scipy.sparse as sp
    J = sp.csr_matrix((out["crs_vals"], out["colind"], out["rowptr"]), shape=(m["ndof"],) * 2)
    fd = -(rp - rm) / (2 * eps)  # the vector holds -res.val()
    assert np.abs(J @ v - fd).max() < 1e-7 * np.abs(fd).max()
    named = oracle.assemble_thermal_fields(dim, order, qdeg, nodes, m["lids"], m["offsets"], u,
                                           {"thermal source": funcs["thermal source"], "thermal diffusion": "kappa0 + 0.2*grad(e)[x]^2"},
                                           rowptr=ref["rowptr"], colind=ref["colind"], functions={"kappa0": "1+e*e"})
    assert np.abs(named["crs_vals"] - out["crs_vals"]).max() < 1e-14 * np.abs(out["crs_vals"]).max()
