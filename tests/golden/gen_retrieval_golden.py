"""Fixture generator (build container only): the reference's `has_answer` / `calculate_matches` (DPR validation) on cases made here.
    python tests/golden/gen_retrieval_golden.py  ->  tests/golden/retrieval_ref.json"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import install_import_shims  # noqa: E402

install_import_shims()
sys.modules.setdefault('spacy', types.ModuleType('spacy'))          # imported at module level by tokenizers.py, unused by SimpleTokenizer
from tasks.openqa.dense_retriever.evaluation import qa_validation as ref  # noqa: E402
from tasks.openqa.dense_retriever.evaluation.tokenizers import SimpleTokenizer  # noqa: E402

CASES = [
    (["Paris"], "The capital of France is paris.", "string"),
    (["New York City"], "He moved to New  York\tcity in 1999", "string"),
    (["new york"], "newyork is not it", "string"),
    (["café"], "A small CAFÉ by the river", "string"),
    (["1,000"], "about 1,000 people", "string"),
    (["1000"], "about 1,000 people", "string"),
    (["O'Neil"], "shaquille o'neil played", "string"),
    (["a b", "river"], "the River Seine", "string"),
    ([""], "anything", "string"),
    (["19\\d\\d"], "born in 1987 somewhere", "regex"),
    (["^the (cat|dog)$"], "The Dog", "regex"),
    (["(unclosed"], "text (unclosed", "regex"),
    (["東京"], "首都は東京です", "string"),
]
DOCS = {1: ("the emperor of rome was augustus", "rome"), 2: ("paris is the capital of france", "paris"), 3: ("no answer here", "x"),
        4: ("Augustus ruled first", "augustus")}
QUESTIONS = [(["augustus"], ([3, 1, 4], [0.9, 0.8, 0.7])), (["Paris", "lyon"], ([2, 3, 1], [0.5, 0.4, 0.3])), (["nothing"], ([1, 2, 3], [1.0, 0.9, 0.8]))]


def main():
    tok = SimpleTokenizer()
    out = {"cases": [[a, t, m, bool(ref.has_answer(a, t, tok, m))] for a, t, m in CASES]}
    stats = ref.calculate_matches(DOCS, [q[0] for q in QUESTIONS], [q[1] for q in QUESTIONS], 1, "string")
    out["docs"] = {str(k): list(v) for k, v in DOCS.items()}
    out["questions"] = [[q[0], list(q[1][0]), list(q[1][1])] for q in QUESTIONS]
    out["top_k_hits"] = list(stats.top_k_hits)
    out["questions_doc_hits"] = [list(map(bool, h)) for h in stats.questions_doc_hits]
    json.dump(out, open(os.path.join(HERE, "retrieval_ref.json"), "w"), ensure_ascii=True, indent=0)
    print([c[3] for c in out["cases"]], out["top_k_hits"])


if __name__ == "__main__":
    main()
