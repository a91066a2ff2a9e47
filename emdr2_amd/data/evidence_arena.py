"""Evidence corpus resident in HBM for the device-side evidence fetch + token assembly
(include/emdr2_assembly.h).  Replaces, per training step, the reference's B*k host iterations over
`WikiTitleDocMap.get_neighbour_paragraphs` (tools/inverted_title_index.py:23-38) and the mmap token
stores `passages_map` / `title_map` (megatron/model/emdr2_model.py:401-408, 457-468).

21M passages x ~130 tokens x 2 B is ~5.5 GB: it simply lives in HBM next to the index (288 GB)."""
import ctypes

import numpy as np
import torch

from emdr2_amd import _native


class EvidenceArena(object):
    def __init__(self, passages, titles, title_keys=None, device=None):
        """passages[d-1], titles[d-1]: token id sequences of doc d (1-based ids, psgs_w100 order;
        anything indexable incl. the reference's indexed datasets).  title_keys[d-1]: hashable title identity
        (defaults to the title token tuple); docs sharing it form a title group in ascending id order
        (WikiTitleDocMap.process_wikipedia, tools/inverted_title_index.py:41-67)."""
        n = len(passages)
        if len(titles) != n:
            raise ValueError("one title per passage")
        p_off = np.zeros(n + 1, dtype=np.int64)
        t_off = np.zeros(n + 1, dtype=np.int64)
        for d in range(n):
            p_off[d + 1] = p_off[d] + len(passages[d])
            t_off[d + 1] = t_off[d] + len(titles[d])
        p_tok = np.empty(int(p_off[-1]), dtype=np.uint16)
        t_tok = np.empty(int(t_off[-1]), dtype=np.uint16)
        groups, order = {}, []
        for d in range(n):
            pt, tt = np.asarray(passages[d]), np.asarray(titles[d])
            if (pt.size and (pt.min() < 0 or pt.max() > 65535)) or (tt.size and (tt.min() < 0 or tt.max() > 65535)):
                raise ValueError("token ids must fit uint16")
            p_tok[p_off[d]:p_off[d + 1]] = pt
            t_tok[t_off[d]:t_off[d + 1]] = tt
            key = title_keys[d] if title_keys is not None else tuple(int(x) for x in tt)
            if key not in groups:
                groups[key] = []
                order.append(key)
            groups[key].append(d + 1)
        g_off = np.zeros(len(order) + 1, dtype=np.int64)
        g_docs = np.empty(n, dtype=np.int32)
        doc_group = np.zeros(n + 1, dtype=np.int32)
        doc_pos = np.zeros(n + 1, dtype=np.int32)
        for gi, key in enumerate(order):
            docs = groups[key]
            g_off[gi + 1] = g_off[gi] + len(docs)
            g_docs[g_off[gi]:g_off[gi + 1]] = docs
            for pos, doc in enumerate(docs):
                doc_group[doc], doc_pos[doc] = gi, pos
        self.n_docs = n
        self.host = dict(passage_tokens=p_tok, passage_off=p_off, title_tokens=t_tok, title_off=t_off,
                         group_docs=g_docs, group_off=g_off, doc_group=doc_group, doc_pos=doc_pos)
        self.device = device
        self.dev = None
        self._struct = None

    @classmethod
    def from_indexed(cls, passages_ds, titles_ds, title_keys, device=None):
        """From the reference's memory-mapped evidence datasets (--indexed-evidence-data-path / --indexed-title-data-path,
        emdr2_model.py:401-407: `passages_map[doc_id - 1]`, `title_map[doc_id - 1]`) and the evidence file's title strings
        (`WikiTitleDocMap`, tools/inverted_title_index.py:40-64).  Token arrays are taken as they lie in the .bin files (no per-document
        Python loop); only the title grouping walks the documents."""
        p_tok, p_off = passages_ds.flat_tokens()
        t_tok, t_off = titles_ds.flat_tokens()
        n = len(p_off) - 1
        if len(t_off) - 1 != n or len(title_keys) != n:
            raise ValueError("one title per passage")
        for a in (p_tok, t_tok):
            if a.size and (int(a.min()) < 0 or int(a.max()) > 65535):
                raise ValueError("token ids must fit uint16")
        self = cls.__new__(cls)
        groups, order = {}, []
        for d, key in enumerate(title_keys):
            g = groups.get(key)
            if g is None:
                g = groups[key] = []
                order.append(key)
            g.append(d + 1)
        g_off = np.zeros(len(order) + 1, dtype=np.int64)
        g_docs = np.empty(n, dtype=np.int32)
        doc_group = np.zeros(n + 1, dtype=np.int32)
        doc_pos = np.zeros(n + 1, dtype=np.int32)
        for gi, key in enumerate(order):
            docs = groups[key]
            g_off[gi + 1] = g_off[gi] + len(docs)
            g_docs[g_off[gi]:g_off[gi + 1]] = docs
            doc_group[docs] = gi
            doc_pos[docs] = np.arange(len(docs), dtype=np.int32)
        self.n_docs = n
        self.host = dict(passage_tokens=np.ascontiguousarray(p_tok, dtype=np.uint16), passage_off=p_off,
                         title_tokens=np.ascontiguousarray(t_tok, dtype=np.uint16), title_off=t_off,
                         group_docs=g_docs, group_off=g_off, doc_group=doc_group, doc_pos=doc_pos)
        self.device, self.dev, self._struct = device, None, None
        return self

    @classmethod
    def synthetic(cls, n_docs, seed=1234, vocab=30522, device=None):
        """Synthetic corpus built directly in HBM for benchmarks (passages U[100,160] tokens, titles U[2,8],
        title groups of 1-10 consecutive ids).  No host copy: only `assemble` works on it."""
        self = cls.__new__(cls)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        g = torch.Generator(device=dev).manual_seed(seed)
        plen = torch.randint(100, 161, (n_docs,), generator=g, device=dev)
        gsize = torch.randint(1, 11, (n_docs,), generator=g, device=dev)              # more than enough groups
        gend = torch.cumsum(gsize, 0)
        n_groups = int(torch.searchsorted(gend, torch.tensor([n_docs], device=dev)).item()) + 1
        gend = torch.clamp(gend[:n_groups], max=n_docs)
        g_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), gend])
        doc = torch.arange(n_docs, device=dev)
        doc_group = torch.searchsorted(gend, doc, right=True).to(torch.int32)
        doc_pos = (doc - g_off[doc_group.long()]).to(torch.int32)
        tlen_g = torch.randint(2, 9, (n_groups,), generator=g, device=dev)
        tlen = tlen_g[doc_group.long()]
        p_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(plen, 0)])
        t_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(tlen, 0)])
        p_tok = torch.randint(5, vocab, (int(p_off[-1]),), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
        t_tok = torch.randint(5, vocab, (int(t_off[-1]),), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
        pad1 = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_docs, self.host, self.device = n_docs, None, dev
        self.dev = dict(passage_tokens=p_tok, passage_off=p_off, title_tokens=t_tok, title_off=t_off,
                        group_docs=(doc + 1).to(torch.int32), group_off=g_off, doc_group=torch.cat([pad1, doc_group]),
                        doc_pos=torch.cat([pad1, doc_pos]))
        st = _native.EvidenceArenaStruct()
        for k, v in self.dev.items():
            setattr(st, k, v.data_ptr())
        st.n_docs = n_docs
        self._struct = st
        return self

    # ---- host views (compat path: the reference's `topk_data` structure) -------------------------------
    def passage(self, doc_id):
        h = self.host
        return h["passage_tokens"][h["passage_off"][doc_id - 1]:h["passage_off"][doc_id]].astype(np.int64).tolist()

    def title(self, doc_id):
        h = self.host
        return h["title_tokens"][h["title_off"][doc_id - 1]:h["title_off"][doc_id]].astype(np.int64).tolist()

    def neighbour_paragraphs(self, doc_id):
        """(doc ids, main_doc_idx) exactly as WikiTitleDocMap.get_neighbour_paragraphs."""
        h = self.host
        g, i = int(h["doc_group"][doc_id]), int(h["doc_pos"][doc_id])
        row = h["group_docs"][h["group_off"][g]:h["group_off"][g + 1]].tolist()
        if i == 0:
            return row[i:i + 3], 0
        if i == len(row) - 1:
            return row[i - 2:i + 1], -1
        return row[i - 1:i + 2], 1

    # ---- device residency --------------------------------------------------------------------------------
    def to_device(self, device=None):
        if not torch.cuda.is_available():
            raise _native.NativeError("EvidenceArena.to_device needs a GPU; there is no CPU fallback")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dev = {}
        for k, v in self.host.items():
            if not v.flags.writeable:
                v = v.copy()                                   # memory-mapped dataset views are read-only
            t = torch.from_numpy(v.view(np.int16) if v.dtype == np.uint16 else v)
            self.dev[k] = t.to(dev)
        s = _native.EvidenceArenaStruct()
        for k in self.host:
            setattr(s, k, self.dev[k].data_ptr())
        s.n_docs = self.n_docs
        self._struct = s
        return self

    def assemble(self, topk_ids, topk, query_uid, query_ids_t5, query_ids_t5_len, seq_length_ret, seq_length,
                 cls_id, sep_id, pad_id):
        """Device tensors: (context_ids [B,K,S_ret], context_types, query_extended [B*K,S], query_single [B*K,S],
        kept_ids [B,K]) -- the outputs of the reference's `postprocess` (emdr2_model.py:250-303)."""
        if self._struct is None:
            self.to_device()
        lib = _native.lib()
        ids = topk_ids.to(torch.int32).contiguous()
        b, kr = ids.shape
        dev = ids.device
        uid = query_uid.to(device=dev, dtype=torch.int64).contiguous()
        q = query_ids_t5.to(device=dev, dtype=torch.int64).contiguous()
        ql = query_ids_t5_len.to(device=dev, dtype=torch.int64).contiguous()
        ctx = torch.empty((b, topk, seq_length_ret), dtype=torch.int64, device=dev)
        typ = torch.empty_like(ctx)
        ext = torch.empty((b * topk, seq_length), dtype=torch.int64, device=dev)
        one = torch.empty_like(ext)
        kept = torch.empty((b, topk), dtype=torch.int32, device=dev)
        _native.check(lib.emdr2_assemble_evidence(ctypes.byref(self._struct), ids.data_ptr(), b, kr, topk, uid.data_ptr(),
                                                  q.data_ptr(), q.shape[1], ql.data_ptr(), seq_length_ret, seq_length,
                                                  cls_id, sep_id, pad_id, ctx.data_ptr(), typ.data_ptr(), ext.data_ptr(),
                                                  one.data_ptr(), kept.data_ptr(), _native.stream_ptr()), "assemble_evidence")
        return ctx, typ, ext, one, kept
