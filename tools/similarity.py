#!/usr/bin/env python3
"""
Token-run similarity of a Python file of this repository against a file it might resemble:
the fraction of the file's tokens (comments, docstrings and layout removed) that lie inside a
run of >= RUN consecutive tokens which also occurs in the other file. Development aid for
keeping this repository's host code its own (python tools/similarity.py A B [RUN]).
"""
import io
import sys
import tokenize


def tokens(path: str) -> list[str]:
    src = open(path, encoding="utf-8").read()
    out, prev = [], None
    skip = {tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING, tokenize.ENDMARKER}
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type in skip:
            prev = tok.type if tok.type in (tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT) else prev
            continue
        # a string that is a statement of its own = docstring
        if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT):
            prev = tokenize.STRING
            continue
        out.append(tok.string)
        prev = tok.type
    return out


def covered_fraction(a: list[str], b: list[str], run: int) -> float:
    grams = set()
    for i in range(len(b) - run + 1):
        grams.add(tuple(b[i:i + run]))
    mark = [False] * len(a)
    for i in range(len(a) - run + 1):
        if tuple(a[i:i + run]) in grams:
            for k in range(i, i + run):
                mark[k] = True
    return sum(mark) / max(len(a), 1)


if __name__ == "__main__":
    run = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    ta, tb = tokens(sys.argv[1]), tokens(sys.argv[2])
    print(f"{covered_fraction(ta, tb, run):.3f}  ({len(ta)} tokens, run >= {run})  {sys.argv[1]}  vs  {sys.argv[2]}")


def show_runs(a_path: str, b_path: str, run: int = 12) -> None:
    """prints the shared runs (development aid)"""
    a, b = tokens(a_path), tokens(b_path)
    grams = {tuple(b[i:i + run]) for i in range(len(b) - run + 1)}
    i = 0
    while i < len(a) - run + 1:
        if tuple(a[i:i + run]) in grams:
            j = i + run
            while j < len(a) and tuple(a[j - run + 1:j + 1]) in grams:
                j += 1
            print(f"[{j - i:4d}] " + " ".join(a[i:j])[:300])
            i = j
        else:
            i += 1
