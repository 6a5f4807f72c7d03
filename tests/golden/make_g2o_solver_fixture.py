"""Extracts the numbers of g2o's known-answer linear system into a JSON fixture.

Source of the DATA (not code): the string/number literals of
/root/reference/third_party/g2o/unit_test/solver/sparse_system_helper.cpp
  :52-149  block-sparse SPD matrix (upper-triangular blocks, 12 x 12 blocks of 3 x 3)
  :151-194 dense inverse of that matrix (36 x 36), used by the reference's SolvePattern / SolveBlocks tests
           (linear_solver_test.cpp:88-140)
  :255     right-hand side b (createTestVectorB)
  :298     expected solution x (createTestVectorX)
checked there by linear_solver_test.cpp:72-85 with isApprox(1e-6).
Run in the build container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_g2o_solver_fixture.py
"""
import json
import os
import re

SRC = "/root/reference/third_party/g2o/unit_test/solver/sparse_system_helper.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g2o_sparse_system.json")


def main():
    txt = open(SRC).read()
    body = txt[txt.index("std::string sparseMatrixString()"):txt.index("std::string denseInverseMatrixString()")]
    lines = re.findall(r'aux << "([^"]*)"', body)
    rbi = [int(v) for v in lines[0].split(":")[1].split()][1:]
    blocks = []
    i = 2
    while i < len(lines):
        m = re.match(r"BLOCK : (\d+) (\d+)", lines[i])
        assert m, lines[i]
        r, c = int(m.group(1)), int(m.group(2))
        rows = [[float(v) for v in lines[i + 1 + k].split()] for k in range(3)]
        blocks.append(dict(r=r, c=c, v=rows))
        i += 4

    def vec(name):
        seg = txt[txt.index("g2o::VectorX %s()" % name):]
        seg = seg[:seg.index("return result;")]
        return [float(v) for v in re.findall(r"result\(idx\+\+\) = ([-0-9.eE+]+);", seg)]

    b, x = vec("createTestVectorB"), vec("createTestVectorX")
    assert len(b) == 36 and len(x) == 36 and rbi[-1] == 36
    ibody = txt[txt.index("std::string denseInverseMatrixString()"):]
    ibody = ibody[:ibody.index("return aux.str();")]
    inv = [[float(v) for v in ln.split()] for ln in re.findall(r'aux << "([^"#]*)"', ibody) if ln.strip()]
    assert len(inv) == 36 and all(len(r) == 36 for r in inv)
    json.dump(dict(source=SRC + ":52-149,151-194,255,298", block_offsets=rbi, blocks=blocks, b=b, x=x, inverse=inv, tol=1e-6),
              open(OUT, "w"), indent=0)
    print("wrote", OUT, len(blocks), "blocks")


if __name__ == "__main__":
    main()
