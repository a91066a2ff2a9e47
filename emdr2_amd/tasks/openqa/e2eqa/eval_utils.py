"""Exact-match metric of the QA task (reference: tasks/openqa/e2eqa/eval_utils.py:23-63): NFD-normalise, lower-case, drop ASCII
punctuation and the articles a/an/the, collapse whitespace; a prediction scores 1 if it equals any reference answer after that."""
import re
import string
import unicodedata

_PUNCT = set(string.punctuation)
_ARTICLES = re.compile(r"\b(a|an|the)\b")


def normalize_answer(s):
    s = unicodedata.normalize("NFD", s).lower()
    s = "".join(ch for ch in s if ch not in _PUNCT)
    return " ".join(_ARTICLES.sub(" ", s).split())


def exact_match_score(prediction, ground_truth):
    return normalize_answer(prediction) == normalize_answer(ground_truth)


def regex_match_score(prediction, ground_truth):
    try:
        return re.compile(ground_truth, flags=re.IGNORECASE + re.UNICODE + re.MULTILINE).match(prediction) is not None
    except re.error:
        return False


def metric_max_over_ground_truths(metric_fn, prediction, ground_truths):
    return max(metric_fn(prediction, gt) for gt in ground_truths)
