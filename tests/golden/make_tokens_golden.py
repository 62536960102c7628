"""Golden table for the input-format DSL: the reference's own ``parse_input_string`` (READ/gl/dataset.py:39-82) executed on a
list of tokens, and its ``generate_input_string`` (:85-122) executed on every parsed configuration and on the configurations of
the reference's own ``test_generate_parse`` (:124-200, the one test the reference holds for this path).  The module imports glumpy (not installed), so the function's SOURCE TEXT is extracted with ``ast`` and
executed against a stub ``NNScene`` that carries the constants read from READ/gl/programs.py:61-75.

    python tests/golden/make_tokens_golden.py          (needs /root/reference; writes tests/golden/input_tokens.json)
"""
import ast
import json
import os
import re

REF = "/root/reference"
TOKENS = ["uv_1d_p1", "uv_1d_p1_ds1", "uv_1d_p1_ds4", "uv_1d_p3", "uv_1d_ps20_ds2", "uv_1d", "uv_2d", "uv_2d_p1",
          "colors_p1", "colors_p2_ds1", "colors", "normals_m_p1", "normals_r_p1_ds1", "normals_l_p4", "normals_d_ps8",
          "xyz_p1", "depth_p1_ds2", "labels_p1", "uv_1d_p12_ds5"]
BAD = ["foo_p1", "p1_uv_1d", ""]


def main():
    src = open(os.path.join(REF, "READ/gl/dataset.py")).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef)
           and n.name in ("parse_input_string", "generate_input_string", "test_generate_parse")]
    consts = dict(re.findall(r"^\s+((?:MODE|UV_TYPE|NORMALS_MODE)_[A-Z0-9_]+)\s*=\s*(\d+)", open(os.path.join(REF, "READ/gl/programs.py")).read(), re.M))
    NNScene = type("NNScene", (), {k: int(v) for k, v in consts.items()})
    env = {"re": re, "NNScene": NNScene}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "dataset.py", "exec"), env)
    table, generated = {}, {}
    for t in TOKENS:
        cfg = env["parse_input_string"](t)
        generated[t] = env["generate_input_string"](cfg)       # the reference's inverse on its own parse result
        cfg["mode"] = list(cfg["mode"])
        table[t] = cfg
    # the reference's own test: its configurations, and what its generate / parse make of them
    seen = []
    real_generate = env["generate_input_string"]

    def spy(config):
        s = real_generate(config)
        seen.append({"config": {**config, "mode": list(config["mode"])}, "string": s})
        return s
    env["generate_input_string"] = spy
    env["test_generate_parse"]()                               # asserts parse(generate(c)) == c inside the reference
    bad = []
    for t in BAD:
        try:
            env["parse_input_string"](t)
        except ValueError:
            bad.append(t)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "input_tokens.json")
    json.dump({"tokens": table, "value_error": bad, "generated": generated, "reference_test_generate_parse": seen},
              open(out, "w"), indent=1, sort_keys=True)
    print(out, len(table), "tokens,", len(bad), "rejected")


if __name__ == "__main__":
    main()
