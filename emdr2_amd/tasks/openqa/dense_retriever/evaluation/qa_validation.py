"""Answer-presence validation of retrieved passages (reference: tasks/openqa/dense_retriever/evaluation/qa_validation.py:29-125 and the
`SimpleTokenizer` of tokenizers.py:150-192 -- the DPR recipe): a passage is a hit if, after NFD normalisation, the uncased token sequence of
any reference answer occurs in the passage's token sequence (match_type 'string'), or the answer regex matches (match_type 'regex').
Tokens: maximal runs of letters / digits / marks, or any single non-space non-control character.  top_k_hits[i] = number of questions
with a hit among their first i+1 passages.  Pinned by tests/golden/retrieval_ref.json (the reference's functions run in the build
container)."""
import collections
import re
import unicodedata

import regex

QAMatchStats = collections.namedtuple('QAMatchStats', ['top_k_hits', 'questions_doc_hits'])
_TOKEN = regex.compile(r'([\p{L}\p{N}\p{M}]+)|([^\p{Z}\p{C}])', flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)


def _normalize(text):
    return unicodedata.normalize('NFD', text)


def words_uncased(text):
    return [m.group().lower() for m in _TOKEN.finditer(text)]


def regex_match(text, pattern):
    try:
        pattern = re.compile(pattern, flags=re.IGNORECASE + re.UNICODE + re.MULTILINE)
    except BaseException:
        return False
    return pattern.search(text) is not None


def has_answer(answers, text, match_type='string'):
    text = _normalize(text)
    if match_type == 'string':
        words = words_uncased(text)
        for answer in answers:
            a = words_uncased(_normalize(answer))
            for i in range(0, len(words) - len(a) + 1):
                if a == words[i:i + len(a)]:
                    return True
    elif match_type == 'regex':
        for answer in answers:
            if regex_match(text, _normalize(answer)):
                return True
    return False


def check_answer(answers, doc_ids, all_docs, match_type):
    hits = []
    for doc_id in doc_ids:
        doc = all_docs.get(doc_id)
        hits.append(bool(doc is not None and doc[0] is not None and has_answer(answers, doc[0], match_type)))
    return hits


def calculate_matches(all_docs, answers, closest_docs, workers_num=1, match_type='string'):
    """all_docs: {doc_id: (text, title)}; answers: one list of strings per question; closest_docs: [(doc_ids, scores)] per question.
    (The reference forks `workers_num` processes; the result is order-independent, so this runs in-process.)"""
    scores = [check_answer(a, ids, all_docs, match_type) for a, (ids, _) in zip(answers, closest_docs)]
    n_docs = len(closest_docs[0][0]) if closest_docs else 0
    top_k_hits = [0] * n_docs
    for question_hits in scores:
        best_hit = next((i for i, x in enumerate(question_hits) if x), None)
        if best_hit is not None:
            top_k_hits[best_hit:] = [v + 1 for v in top_k_hits[best_hit:]]
    return QAMatchStats(top_k_hits, scores)
