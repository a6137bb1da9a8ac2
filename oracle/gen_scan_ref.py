#!/usr/bin/env python3
"""oracle/gen_scan_ref.py — TEST INFRASTRUCTURE.  Cuts the reference's exact-scan code out of the sources WHERE THEY LIE
(/root/reference) into include fragments under oracle/_ref/ (git-ignored: no reference source ever enters this
repository's history) so that oracle/scan_ref_wrap.cpp can compile the reference's OWN loop as a checker:

  sqlite_vec_backend.cpp   enum class ExactRowSelection                        -> scan_ref_file_scope.inc
                           isZeroNormEmbedding .. isFiniteEmbedding (:204-236)    (verbatim)
                           kCreateVectorsTable (:342-371)
                           Impl::recordFromStatement (:3102-3160)              -> scan_ref_members.inc (verbatim)
                           Impl::bruteForceSearchUnlocked (:4114-4409)
  vector_database.cpp      VectorDatabase::computeCosineSimilarity (:1786-1810) -> scan_ref_cosine.inc (verbatim)
  sqlite_vec_backend.cpp   StmtResetGuard (:319-328), kSelectByRowid (:440-447)  -> scan_ref_vec0_file_scope.inc (verbatim)
  (round 6: the in-tree      Impl::vec0TableName (:617-619), getVectorByRowidUnlocked (:3084-3099), ensureVec0TableUnlocked
   half of the L2 path)    (:3236-3249), decodeVectorForDimRowUnlocked (:3251-3266), rebuildVec0DimUnlocked (:3350-3421),
                           vec0SearchUnlocked (:4450-4530)                      -> scan_ref_vec0_members.inc (verbatim)

Fragments are located by their opening text and closed by brace matching (strings, raw strings and comments skipped), so
a reference that moves a few lines still extracts; the line spans found are written into the fragments' headers and
checked against the spans SURVEY.md 8 cites (a warning when they drift, an error when a fragment is not found).

    python oracle/gen_scan_ref.py [--ref /root/reference] [--out oracle/_ref]
"""
from __future__ import annotations

import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def _skip_to_block_end(text: str, start: int) -> int:
    """Index just past the `}` that closes the first `{` at or after `start`; C++ comments, string / char literals and
    raw strings are skipped."""
    i, n, depth, seen = start, len(text), 0, False
    while i < n:
        c = text[i]
        two = text[i:i + 2]
        if two == "//":
            i = text.index("\n", i)
            continue
        if two == "/*":
            i = text.index("*/", i) + 2
            continue
        if c == "R" and text[i + 1] == '"':                       # raw string R"delim( ... )delim"
            p = text.index("(", i)
            delim = text[i + 2:p]
            i = text.index(")" + delim + '"', p) + len(delim) + 2
            continue
        if c == '"' or c == "'":
            q = c
            i += 1
            while text[i] != q:
                i += 2 if text[i] == "\\" else 1
            i += 1
            continue
        if c == "{":
            depth += 1
            seen = True
        elif c == "}":
            depth -= 1
            if seen and depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced braces")


def _line_of(text: str, pos: int) -> int:
    return text.count("\n", 0, pos) + 1


def cut(text: str, opening: str, expect: tuple[int, int], what: str, trailing: str = "", to_statement_end: bool = False,
        lines_above: int = 0):
    """(fragment, first line, last line): from the start of the line holding `opening` to the end of its block
    (+ `trailing`, e.g. the `;` of an enum).  to_statement_end: the fragment ends at the first `;` after the opening
    at brace depth 0 (a constant initialised with a raw string)."""
    at = text.find(opening)
    if at < 0:
        raise SystemExit(f"gen_scan_ref: '{what}' not found in the reference (opening text: {opening!r})")
    if text.find(opening, at + 1) >= 0:
        raise SystemExit(f"gen_scan_ref: '{what}' is ambiguous in the reference")
    begin = text.rfind("\n", 0, at) + 1
    for _ in range(lines_above):                 # (a return type that has a line of its own above the name)
        begin = text.rfind("\n", 0, begin - 1) + 1
    if to_statement_end:
        i = at
        while True:
            if text[i] == "R" and text[i + 1] == '"':
                p = text.index("(", i)
                delim = text[i + 2:p]
                i = text.index(")" + delim + '"', p) + len(delim) + 2
                continue
            if text[i] == ";":
                end = i + 1
                break
            i += 1
    else:
        end = _skip_to_block_end(text, at)
        if trailing:
            if text[end:end + len(trailing)] != trailing:
                raise SystemExit(f"gen_scan_ref: '{what}' does not end in {trailing!r}")
            end += len(trailing)
    lo, hi = _line_of(text, begin), _line_of(text, end - 1)
    if abs(lo - expect[0]) > 40 or abs(hi - expect[1]) > 40:
        sys.stderr.write(f"gen_scan_ref: warning: '{what}' found at :{lo}-{hi}, SURVEY.md cites :{expect[0]}-{expect[1]}\n")
    return text[begin:end] + "\n", lo, hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "_ref"))
    a = ap.parse_args()
    backend_path = os.path.join(a.ref, "src", "vector", "sqlite_vec_backend.cpp")
    vdb_path = os.path.join(a.ref, "src", "vector", "vector_database.cpp")
    backend = open(backend_path, encoding="utf-8").read()
    vdb = open(vdb_path, encoding="utf-8").read()
    os.makedirs(a.out, exist_ok=True)

    def banner(path, spans):
        return ("// GENERATED by oracle/gen_scan_ref.py from " + path + " — verbatim reference text, cut where it lies; NOT part of\n"
                "// this repository (oracle/_ref/ is git-ignored).  Spans: " + ", ".join(f"{w} :{lo}-{hi}" for w, lo, hi in spans) + "\n")

    enum_, e0, e1 = cut(backend, "enum class ExactRowSelection {", (76, 79), "ExactRowSelection", trailing=";")
    h_zero, z0, _ = cut(backend, "inline bool isZeroNormEmbedding(", (204, 211), "isZeroNormEmbedding")
    h_fin, _, f1 = cut(backend, "inline bool isFiniteEmbedding(", (229, 236), "isFiniteEmbedding")
    schema, s0, s1 = cut(backend, "constexpr const char* kCreateVectorsTable = R\"sql(", (342, 371), "kCreateVectorsTable",
                         to_statement_end=True)
    spans = [("ExactRowSelection", e0, e1), ("isZeroNormEmbedding", z0, z0 + h_zero.count("\n") - 1),
             ("isFiniteEmbedding", f1 - h_fin.count("\n") + 1, f1), ("kCreateVectorsTable", s0, s1)]
    with open(os.path.join(a.out, "scan_ref_file_scope.inc"), "w") as f:
        f.write(banner(backend_path, spans) + enum_ + "\n" + h_zero + "\n" + h_fin + "\n" + schema)

    rec, r0, r1 = cut(backend, "VectorRecord recordFromStatement(sqlite3_stmt* stmt) const {", (3102, 3160), "recordFromStatement")
    # the function's declaration starts one line above its name (the return type has a line of its own)
    at = backend.find("    bruteForceSearchUnlocked(const std::vector<float>& query_embedding, size_t k,")
    if at < 0:
        raise SystemExit("gen_scan_ref: bruteForceSearchUnlocked not found")
    prev = backend.rfind("\n", 0, at - 1) + 1
    ret_line = backend[prev:at]
    if "Result<std::vector<VectorRecord>>" not in ret_line:
        raise SystemExit("gen_scan_ref: unexpected return-type line above bruteForceSearchUnlocked: " + ret_line.strip())
    end = _skip_to_block_end(backend, at)
    scan = backend[prev:end] + "\n"
    b0, b1 = _line_of(backend, prev), _line_of(backend, end - 1)
    if abs(b0 - 4114) > 60 or abs(b1 - 4409) > 60:
        sys.stderr.write(f"gen_scan_ref: warning: bruteForceSearchUnlocked found at :{b0}-{b1}, SURVEY.md cites :4115-4409\n")
    with open(os.path.join(a.out, "scan_ref_members.inc"), "w") as f:
        f.write(banner(backend_path, [("recordFromStatement", r0, r1), ("bruteForceSearchUnlocked", b0, b1)]) + rec + "\n" + scan)

    cos, c0, c1 = cut(vdb, "double VectorDatabase::computeCosineSimilarity(", (1786, 1810), "computeCosineSimilarity")
    with open(os.path.join(a.out, "scan_ref_cosine.inc"), "w") as f:
        f.write(banner(vdb_path, [("computeCosineSimilarity", c0, c1)]) + cos)
    # ---- round 6: the in-tree half of the L2 (vec0) path --------------------------------------------------------------------
    guard, g0, g1 = cut(backend, "struct StmtResetGuard {", (319, 328), "StmtResetGuard", trailing=";")
    sel, q0, q1 = cut(backend, "constexpr const char* kSelectByRowid = R\"sql(", (440, 447), "kSelectByRowid", to_statement_end=True)
    with open(os.path.join(a.out, "scan_ref_vec0_file_scope.inc"), "w") as f:
        f.write(banner(backend_path, [("StmtResetGuard", g0, g1), ("kSelectByRowid", q0, q1)]) + guard + "\n" + sel)
    vparts = [
        cut(backend, "std::string vec0TableName(size_t dim) const {", (617, 619), "vec0TableName"),
        cut(backend, "std::optional<VectorRecord> getVectorByRowidUnlocked(int64_t rowid) const {", (3084, 3099), "getVectorByRowidUnlocked"),
        cut(backend, "Result<void> ensureVec0TableUnlocked(size_t dim) {", (3236, 3249), "ensureVec0TableUnlocked"),
        cut(backend, "decodeVectorForDimRowUnlocked(sqlite3_stmt* stmt, size_t dim) const {", (3250, 3266), "decodeVectorForDimRowUnlocked",
            lines_above=1),
        cut(backend, "Result<void> rebuildVec0DimUnlocked(size_t dim) {", (3350, 3421), "rebuildVec0DimUnlocked"),
        cut(backend, "    vec0SearchUnlocked(const std::vector<float>& query_embedding, size_t k,", (4450, 4530), "vec0SearchUnlocked",
            lines_above=1),
    ]
    vnames = ["vec0TableName", "getVectorByRowidUnlocked", "ensureVec0TableUnlocked", "decodeVectorForDimRowUnlocked",
              "rebuildVec0DimUnlocked", "vec0SearchUnlocked"]
    vspans = [(n_, lo, hi) for n_, (_, lo, hi) in zip(vnames, vparts)]
    if "Result<std::vector<VectorRecord>>" not in vparts[-1][0].splitlines()[0]:
        raise SystemExit("gen_scan_ref: unexpected return-type line above vec0SearchUnlocked: " + vparts[-1][0].splitlines()[0].strip())
    with open(os.path.join(a.out, "scan_ref_vec0_members.inc"), "w") as f:
        f.write(banner(backend_path, vspans) + "\n".join(t for t, _, _ in vparts))
    with open(os.path.join(a.out, "scan_ref_spans.txt"), "w") as f:
        for w, lo, hi in spans + [("recordFromStatement", r0, r1), ("bruteForceSearchUnlocked", b0, b1),
                                  ("StmtResetGuard", g0, g1), ("kSelectByRowid", q0, q1)] + vspans:
            f.write(f"src/vector/sqlite_vec_backend.cpp:{lo}-{hi} {w}\n")
        f.write(f"src/vector/vector_database.cpp:{c0}-{c1} computeCosineSimilarity\n")
    print("gen_scan_ref: fragments written to", a.out)


if __name__ == "__main__":
    main()
