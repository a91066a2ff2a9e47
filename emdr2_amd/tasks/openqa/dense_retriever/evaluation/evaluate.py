"""Retrieval-accuracy evaluation of the dual encoder (reference: tasks/openqa/dense_retriever/evaluation/evaluate.py:17-170,
`OpenRetrievalEvaluator`): embed the questions with the query tower, search the evidence index FAISS-style (fp32 scores, top-k 100 by
default), and report the fraction of questions with an answer-bearing passage among the first k for each --report-topk-accuracies k.

MI355X-native differences: the index is the row-sharded in-HBM `FaissMIPSIndex` (every rank searches its shard for all questions and
the results are merged on the device, instead of rank 0 searching and broadcasting, evaluate.py:100-127)."""
import csv

import torch

from emdr2_amd.data.emdr2_index import FaissMIPSIndex
from emdr2_amd.tasks.openqa.dense_retriever.evaluation.qa_validation import calculate_matches


def read_evidence_text(evidence_data_path):
    """{doc_id: (text, title)} from the evidence TSV (doc_id, text, title; one header line) -- orqa_wiki_dataset.py:176-205."""
    docs = {}
    with open(evidence_data_path) as f:
        reader = csv.reader(f, delimiter='\t')
        next(reader, None)
        for row in reader:
            docs[int(row[0])] = (row[1], row[2])
    return docs


def read_qa_file(qa_file):
    """[(question, [answers])] from a `question <TAB> ["answer", ...]` file (evaluation/data.py)."""
    from ast import literal_eval
    out = []
    with open(qa_file) as f:
        for row in csv.reader(f, delimiter='\t'):
            out.append((row[0], literal_eval(row[1])))
    return out


class OpenRetrievalEvaluator(object):
    def __init__(self, query_model, tokenizer, embed_data, all_docs, hidden_size, seq_length_ret=256, batch_size=128, topk_retrievals=100,
                 report_topk_accuracies=(1, 5, 20, 100), match='string', process_group=None):
        """query_model: `PretrainedBertModel`; embed_data: `OpenRetreivalDataStore` (or None + `mips_index.add_flat_file` afterwards)."""
        self.model, self.tokenizer, self.all_docs = query_model, tokenizer, all_docs
        self.seq_length_ret, self.batch_size, self.topk, self.report, self.match = seq_length_ret, batch_size, topk_retrievals, report_topk_accuracies, match
        self.mips_index = FaissMIPSIndex(embed_size=hidden_size, embed_data=embed_data, use_gpu=True, process_group=process_group)

    def generate_query_vectors(self, questions):
        """[CLS] q [SEP] pad -> hidden state of [CLS] (evaluate.py:55-92), eval mode."""
        from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import build_tokens_types_paddings_from_ids
        t = self.tokenizer
        was_training = self.model.training
        self.model.eval()
        out = []
        try:
            with torch.no_grad():
                for lo in range(0, len(questions), self.batch_size):
                    ids = [build_tokens_types_paddings_from_ids(t.tokenize(q), [], self.seq_length_ret, 2, t.cls, t.sep, t.pad, t.bos_token_id,
                                                                t.eos_token_id)[0] for q in questions[lo:lo + self.batch_size]]
                    ids = torch.tensor(ids, dtype=torch.int64, device="cuda")
                    out.append(self.model(ids, torch.zeros_like(ids)))
        finally:
            self.model.train(was_training)
        return torch.cat(out, dim=0)

    def evaluate(self, qa_file, split="test"):
        qa = read_qa_file(qa_file)
        q = self.generate_query_vectors([x[0] for x in qa])
        distance, topkindex = self.mips_index.search_mips_index(q, top_k=self.topk, reconstruct=False)
        closest = [(topkindex[i].tolist(), distance[i].tolist()) for i in range(len(qa))]
        stats = calculate_matches(self.all_docs, [x[1] for x in qa], closest, match_type=self.match)
        acc = {k: stats.top_k_hits[k - 1] / max(len(qa), 1) for k in self.report if k <= self.topk}
        if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
            print("{} SET RESULTS".format(split), flush=True)
            for k, v in acc.items():
                print("top-{}: {:.2f}".format(k, v * 100), flush=True)
        return acc, stats
