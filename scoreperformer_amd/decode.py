"""Greedy render engine (b = 1): static caches, device-resident position, one hipGraph-captured decoder step per note.

Semantics = `ScorePerformerMixedLMWrapper.unmask_tokens` with `filter_logits_fn=top_k, k=1` (wrappers.py:325-407): for every
position idx that still holds MASK tokens, run the shifted decoder on the prefix, take the LM-head logits of position idx-1 for
the masked dims, ban PAD/MASK ids, write the arg-max.  Differences in *how*: the reference re-runs Python modules per note with
`torch.cat`-grown caches and reads tokens back to the host every step; here every note is one replay of a captured graph of
fp32 kernels over preallocated [L, .] caches, and tokens stay on the device until the end.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .models.scoreperformer.embeddings import build_tables, TupleTokenTiedLMHead, TupleTokenLMHead
from .modules.layers import AdaptiveLayerNorm
from .modules.transformer.attend import AttentionIntermediates
from .modules.transformer.transformer import TransformerIntermediates

F32 = torch.float32


class GreedyDecoder:
    def __init__(self, decoder, max_len: int, use_graph: bool = True, fused: bool = True, attn_splits: int = 16,
                 reference_compat: bool = False):
        """decoder: the TupleTransformer wrapped by the MixedLM wrapper (`model.perf_decoder.model`).
        fused: ~40 fused launches per note (LayerNorm inside the GEMVs, GLU epilogue, split-key attention, one LM-head launch)
        instead of ~110 small ones; same arithmetic, same tokens.
        reference_compat (cross-attending decoders only; no effect otherwise): reproduce the reference's cached decode TOKEN FOR TOKEN,
        including its defect in this mode -- under the cache protocol a cross-attention block returns one row per prefix position
        (modules/transformer/transformer.py:201, attention.py:216-218), the rows pile up in the cached final hiddens and
        `hidden_state[:, idx - 1]` (wrappers.py:364) reads the hidden of a STALE position from the third decoded note on: decoded note
        number s (1-based) is predicted from the hidden of position first - 1 + (t - 1), t the smallest integer with t (t + 1) / 2 >= s.
        Default False: the evident intent, row idx - 1."""
        self.m = m = decoder
        import os
        attn_splits = int(os.environ.get("SPN_DEC_SPLITS", attn_splits))   # tuning aid
        self.head_slabs = int(os.environ.get("SPN_DEC_HEAD_SLABS", 8))
        self.legacy_launches = os.environ.get("SPN_DEC_LEGACY", "0") == "1"   # A/B aid: round-1 launch list (separate embed / merge kernels)
        self.max_len, self.use_graph, self.fused, self.attn_splits = max_len, use_graph, fused, attn_splits
        # one persistent launch per ('a', 'f') layer pair instead of five (csrc/decode_layer.hip)
        self.use_pair = os.environ.get("SPN_DEC_PAIR", "1") != "0" and not self.legacy_launches
        self.pair_groups = 0       # workgroups of that launch (0: shape not supported, the five launches run)
        self.pair_front = self.pair_tail = False   # the note's input projections / the LM head's input projection ride in that launch
        self.pair_fallbacks = 0    # how often a timed-out hand-off made this engine fall back to the five launches per pair (see _pair_failed)
        self._pair_fault_inject = False   # test hook: start with the error word set, as after a timed-out hand-off
        tr = m.transformer
        types = tuple(tr.layer_types)
        self.cross = 'c' in types
        block = ('a', 'c', 'f') if self.cross else ('a', 'f')
        if types != block * (len(types) // len(block)) or not tr.pre_norm:
            raise NotImplementedError("decode engine: pre-norm decoder with ('a','f') or ('a','c','f') layer blocks")
        if getattr(m.token_emb, "multiseq_mode", None) != "post-cat" or m.pos_emb is not None:
            raise NotImplementedError("decode engine: multi-seq post-cat token embeddings without absolute positions")
        for kind, (_norms, blk, _res) in zip(types, tr.layers):
            # the engine's q | k | v rows, caches and out-projection columns are laid out for 64-wide heads and it knows no learned memory
            # rows; `Attention` serves both on the module path (zero-padded weights, memories in front of the keys), which the wrapper falls
            # back to on this error
            if kind in "ac" and (blk.dim_head != 64 or blk.num_mem_kv > 0):
                raise NotImplementedError("decode engine: attention blocks with dim_head == 64 and num_mem_kv == 0")
        self.dev = next(m.parameters()).device
        self.dim = m.dim
        self.ada = tr.ada_norm
        self.reference_compat = bool(reference_compat) and self.cross
        self.stale_tab = None      # reference_compat: int32 [max_len] device table, position -> row of the final hiddens the head reads
        self.nk_dev = None         # render sessions with a cross-attending decoder: int32 device scalar, valid context rows
        self.graph = None
        self.graph_multi = None
        self.graph_notes = max(1, int(os.environ.get("SPN_DEC_GRAPH_NOTES", "16")))   # notes per graph replay (1: one note per replay)
        # several notes per LAUNCH when the whole note is one persistent launch (spn_dec_pairs_notes); 0: one launch per note
        self.multi_note = os.environ.get("SPN_DEC_MULTI_NOTE", "1") != "0"
        self.sampling = None       # None = arg-max; dict(topk=int32 device tensor [n dims], temperature=float) = top-k sampling

    # -- buffers -------------------------------------------------------------------------------------------
    def _alloc(self, L):
        import os
        m, dev, d = self.m, self.dev, self.dim
        te = m.token_emb
        z = lambda *s: torch.zeros(*s, device=dev, dtype=F32)
        # position scalars: `pos` is what the launches of a step read; `pos_next` is read by the first launch of a step only
        # (ops.dec_step_begin latches it into `pos`) and written by its last one (the head: position + 1) -- no separate "advance" launch
        self.pos2 = torch.zeros(2, device=dev, dtype=torch.int32)
        self.pos, self.pos_next = self.pos2[0:1], self.pos2[1:2]
        self.pos_stale = torch.zeros(1, device=dev, dtype=torch.int32)
        self.x0 = z(d)   # token embedding of the position before the (LN | context | style) projection writes the stream `x`
        self.e_cat = z(te.total_emb_dim)
        self.proj_cat = z(2 * d)
        self.tok_emb = z(L, d)
        width = d + (m.context_emb_dim if m.context_emb_mode == "cat" else 0) + (m.style_emb_dim if m.style_emb_mode == "cat" else 0)
        self.xcat = z(width)
        self.x, self.h, self.o = z(d), z(d), z(d)
        att0 = m.transformer.layers[0][1]
        self.heads, self.kvh = att0.heads, att0.kv_heads
        self.qkv = z((self.heads + 2 * self.kvh) * 64)
        ffn = next(blk for lt, (_n, blk, _r) in zip(m.transformer.layer_types, m.transformer.layers) if lt == 'f')
        inner2 = ffn.ff[0].proj.weight.shape[0] if ffn.glu else ffn.ff[0][0].weight.shape[0]
        self.u = z(inner2)
        self.g = z(inner2 // 2 if ffn.glu else inner2)
        self.gb = z(2 * d)
        n_attn = m.transformer.num_attn_layers
        n_self = sum(1 for t in m.transformer.layer_types if t == 'a')
        self.kc = [z(L, self.kvh * 64) for _ in range(n_self)]
        self.vc = [z(L, self.kvh * 64) for _ in range(n_self)]
        self.hid = [z(L, d) for _ in range(n_self + 1)]
        self.q_only = z(self.heads * 64)                      # query of a cross-attention layer
        self.xk, self.xv, self.xmask = [], [], None            # projected context per 'c' layer: filled by _project_context
        self.e_head = z(te.total_emb_dim)
        self.e_head_n = z(te.total_emb_dim)
        self.logits = z(max(te.num_tokens.values()) + 8)
        self.stats = z(2)
        # fused path: split-attention scratch, running max |k|^2 per layer, all AdaLN (gamma|beta) rows of a step in one GEMV
        S = self.attn_splits
        self.att_part = z(self.heads * S * 66)
        self.att_counter = torch.zeros(self.heads, device=dev, dtype=torch.int32)
        self.head_part = z(16 * 8 * 2)
        self.head_counter = torch.zeros(16, device=dev, dtype=torch.int32)
        self.head_logits = z(16, 1024)
        self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        self.kmax2 = [z(self.kvh) for _ in range(n_self)]
        if self.use_pair:
            self.pair_groups = ops.dec_pair_groups(d, self.heads, self.kvh, self.g.numel(), S)
            if self.pair_groups > torch.cuda.get_device_properties(dev).multi_processor_count:
                # every workgroup of that launch must be resident at once, one per CU.  This compares against the device's CU COUNT only:
                # a CU mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) or another kernel holding CUs is not visible here -- then a hand-off poll
                # runs into its bound, the kernel sets *err, and _pair_failed() falls back to the five launches per pair.
                self.pair_groups = 0
        self.pair_front = self.pair_tail = self.pair_head = self.pair_embed = False
        if self.pair_groups:   # hand-off granules of the persistent layer-pair launch ({epoch, value} words: zero = no epoch)
            zg = lambda n: torch.zeros(n, device=dev, dtype=torch.int64)
            self.pair_g = dict(gq=zg(self.qkv.numel()), gp=zg(self.heads * S * 66), go=zg(self.heads * 64), gx=zg(d), gg=zg(self.g.numel()),
                               gxo=zg(d))
            self.pair_jlo = [torch.zeros(self.heads, device=dev, dtype=torch.int32) for _ in range(n_self)]
            self.pair_g2 = dict(gf=zg(d), gxf=zg(d), ge=zg(2048), gh=zg(16 * 16 * 2), gin=zg(2048), gt=zg(16), gl=zg(16 * 1024))
            self.pair_front = self.pair_tail = self.pair_head = self.pair_embed = False
            self.pair_chains = {}      # first layer index of a chain -> ops.DecPairChain (argument records, host + device copy)
            self.pair_tick = torch.zeros(1, device=dev, dtype=torch.int32)
            self.pair_err = torch.zeros(1, device=dev, dtype=torch.int32)
            if self._pair_fault_inject:
                self.pair_err.fill_(9)
            self.pair_stamps = ([torch.zeros(self.pair_groups * 8, device=dev, dtype=torch.int64) for _ in range(n_self)]
                                if os.environ.get("SPN_DEC_PAIR_STAMPS", "0") == "1" else None)   # tuning aid (tools/bench_dec_pair.py; the
            # kernel records them only in a library built with -DSPN_DEC_STAMPS: tools/build_variant.py decode_layer.hip stamps_spn.so -DSPN_DEC_STAMPS)
        tr = m.transformer
        self.norm_list = [norms[0] for norms, _, _ in tr.layers] + ([tr.final_norm] if not isinstance(tr.final_norm, nn.Identity) else [])
        self.ada_rows = {}
        if self.ada:
            ada = [n for n in self.norm_list if isinstance(n, AdaptiveLayerNorm)]
            if ada:
                self.ada_W = torch.cat([n.linear.weight.data.float() for n in ada], 0).contiguous()
                self.ada_b = torch.cat([n.linear.bias.data.float() for n in ada], 0).contiguous()
                # two row sets: the persistent launch computes the NEXT note's rows while this note's are in use (spn_dec_chain_ext.ada_par);
                # every other path uses set 0
                self.gb_both = z(2, len(ada), 2 * d)
                self.gb_all = self.gb_both[0]
                self.ada_rows = {id(n): i for i, n in enumerate(ada)}
        head = m.lm_head
        if isinstance(head, TupleTokenTiedLMHead) and head.reuse_projection:
            self.head_Wt = head.project_emb.weight.data.float().t().contiguous()   # [total_emb, d]: row-major GEMV operand

    def _slopes(self, block):
        """ALiBi slopes of an attention block as a contiguous fp32 [h] tensor, computed ONCE per engine: the weights do not change during a
        render, and `learned_logslopes.exp()` inside the captured step was six extra launches per note (28 us of 329)."""
        cache = self.__dict__.setdefault("_slope_cache", {})
        key = id(block)
        if key not in cache:
            cache[key] = block.rel_pos.padded_slopes().detach().float().contiguous().clone() if block.rel_pos is not None else None
        return cache[key]

    def _project_context(self, context: torch.Tensor, context_mask: Optional[torch.Tensor]):
        """Keys / values of every cross-attention layer over the WHOLE context, once per render: the reference recomputes them from the
        context at every note (the 'c' block gets no cache: modules/transformer/transformer.py:201), the values are the same."""
        ctx = context.float().contiguous()
        self.xk, self.xv = [], []
        for lt, (_n, block, _r) in zip(self.m.transformer.layer_types, self.m.transformer.layers):
            if lt == 'c':
                self.xk.append(ops.gemm_f32(ctx, block.to_k.weight.data))
                self.xv.append(ops.gemm_f32(ctx, block.to_v.weight.data))
        self.xmask = None
        if context_mask is not None:
            cm = context_mask.reshape(-1).contiguous()
            self.xmask = cm.view(torch.uint8) if cm.dtype == torch.bool else cm.to(torch.uint8)

    def _cross_layer(self, ci: int, norm, block, fused: bool):
        """x += to_out(attend(to_q(norm(x)), K_ctx, V_ctx)) for the current position."""
        if fused:
            mode, g_, b_, eps_ = self._norm_args(norm)
            ops.dec_fused_gemv(block.to_q.weight.data, self.x, self.q_only, norm=mode, gamma=g_, beta=b_, eps=eps_)
        else:
            self._ln(self.x, norm, self.h, True)
            ops.dec_gemv(block.to_q.weight.data, self.h, self.q_only)
        slopes = self._slopes(block)
        ops.dec_xattn(self.q_only, self.xk[ci], self.xv[ci], slopes, self.xmask, None if fused else self.o, self.att_part, self.att_counter,
                      h=self.heads, kvh=self.kvh, scale=block.scale, splits=self.attn_splits, nk_dev=self.nk_dev)
        if fused:   # the output projection merges the split-key partials in its prologue: no fences / last-block tail in the attention kernel
            ops.dec_attn_out(block.to_out.weight.data, self.att_part, self.x, h=self.heads, splits=self.attn_splits, residual=self.x)
        else:
            ops.dec_gemv(block.to_out.weight.data, self.o, self.x, residual=self.x)

    # -- one decoder step at position t = *pos (predicts the MASKed dims of position t + 1) -----------------
    def _ln(self, x, norm, out, cond_row: bool):
        if self.ada and isinstance(norm, AdaptiveLayerNorm):
            ops.dec_gemv(norm.linear.weight.data, self.style2d, self.gb, bias=norm.linear.bias.data, pos=self.pos,
                         x_ld=self.style2d.stride(0), x_off=1)
            ops.layernorm_fwd(x.view(1, -1), None, None, self.gb.view(1, -1), out_dtype=F32, eps=norm.eps, out=out.view(1, -1))
        else:
            ops.layernorm_fwd(x.view(1, -1), norm.weight.data, norm.bias.data, None, out_dtype=F32, eps=norm.eps, out=out.view(1, -1))

    def _step(self, dims: List[int]):
        m, d, pos = self.m, self.dim, self.pos
        te = m.token_emb
        has_norm = isinstance(te.norm, nn.LayerNorm)
        gam, bet, eps = (te.norm.weight.data, te.norm.bias.data, te.norm.eps) if has_norm else (None, None, 1e-5)
        for si, (toks, off) in enumerate(((self.seq2d, 0), (self.masked2d, 1))):
            ops.dec_embed(self.tables, toks, pos, self.e_cat, row_off=off, gamma=gam, beta=bet, eps=eps)
            ops.dec_gemv(te.project_emb.weight.data, self.e_cat, self.proj_cat[si * d:(si + 1) * d], bias=te.project_emb.bias.data)
        ops.dec_gemv(te.project_multiemb.weight.data, self.proj_cat, self.x, bias=te.project_multiemb.bias.data)
        ops.dec_copy_row(self.x, self.tok_emb, pos, d, dst_ld=d)
        if isinstance(m.emb_norm, nn.LayerNorm):
            ops.layernorm_fwd(self.x.view(1, -1), m.emb_norm.weight.data, m.emb_norm.bias.data, None, out_dtype=F32,
                              eps=m.emb_norm.eps, out=self.xcat[:d].view(1, -1))
        else:
            ops.dec_copy_row(self.x, self.xcat, pos, d)
        col = d
        if m.context_emb_mode == "cat" and self.ctx2d is not None:
            ops.dec_copy_row(self.ctx2d, self.xcat[col:], pos, m.context_emb_dim, src_ld=self.ctx2d.stride(0), src_off=1)
            col += m.context_emb_dim
        if m.style_emb_mode == "cat" and self.style2d is not None:
            ops.dec_copy_row(self.style2d, self.xcat[col:], pos, m.style_emb_dim, src_ld=self.style2d.stride(0), src_off=1)
        if isinstance(m.project_emb, nn.Linear):
            ops.dec_gemv(m.project_emb.weight.data, self.xcat, self.x, bias=m.project_emb.bias.data)
        else:
            ops.dec_copy_row(self.xcat, self.x, pos, d)
        ai = ci = 0
        for lt, (norms, block, _res) in zip(m.transformer.layer_types, m.transformer.layers):
            if lt == 'c':
                self._cross_layer(ci, norms[0], block, False)
                ci += 1
            elif lt == 'a':
                ops.dec_copy_row(self.x, self.hid[ai], pos, d, dst_ld=d)
                self._ln(self.x, norms[0], self.h, True)
                wqkv = block._fused("_w_qkv", (block.to_q.weight, block.to_k.weight, block.to_v.weight)).data
                ops.dec_gemv(wqkv, self.h, self.qkv)
                slopes = self._slopes(block)
                ops.dec_attn(self.qkv, self.kc[ai], self.vc[ai], slopes, pos, self.o, h=self.heads, kvh=self.kvh, scale=block.scale)
                ops.dec_gemv(block.to_out.weight.data, self.o, self.x, residual=self.x)
                ai += 1
            else:
                self._ln(self.x, norms[0], self.h, True)
                if block.glu:
                    lin = block.ff[0].proj
                else:
                    lin = block.ff[0][0]
                ops.dec_gemv(lin.weight.data, self.h, self.u, bias=lin.bias.data if lin.bias is not None else None)
                ops.dec_glu(self.u, self.g, self.g.numel(), act=block.act_code, glu=block.glu)
                if isinstance(block.ff[1], nn.LayerNorm):
                    ops.layernorm_fwd(self.g.view(1, -1), block.ff[1].weight.data, block.ff[1].bias.data, None, out_dtype=F32,
                                      eps=block.ff[1].eps, out=self.g.view(1, -1))
                out = block.ff[3]
                ops.dec_gemv(out.weight.data, self.g, self.x, bias=out.bias.data if out.bias is not None else None, residual=self.x)
        fn = m.transformer.final_norm
        if not isinstance(fn, nn.Identity):
            self._ln(self.x, fn, self.h, True)
        else:
            ops.dec_copy_row(self.x, self.h, pos, d)
        ops.dec_copy_row(self.h, self.hid[-1], pos, d, dst_ld=d)
        self._head_tail(dims)

    def _head_tail(self, dims: List[int]):
        m, pos = self.m, self.pos
        te = m.token_emb
        # LM head on this position for the candidate dims, arg-max written where the next position holds MASK
        head = m.lm_head
        keys = list(te.embs.keys())
        if self.reference_compat:   # the head reads the cached final hidden of the (stale) row the reference reads: see __init__
            ops.dec_lookup(self.stale_tab, pos, self.pos_stale)
            ops.dec_copy_row(self.hid[-1], self.h, self.pos_stale, self.dim, src_ld=self.dim, src_off=0)
        if isinstance(head, TupleTokenTiedLMHead):
            if head.reuse_projection:
                ops.dec_gemv(head.project_emb.weight.data, self.h, self.e_head, kn_layout=True)
            else:
                ops.dec_gemv(head.project_emb.weight.data, self.h, self.e_head)
            ops.layernorm_fwd(self.e_head.view(1, -1), head.norm.weight.data, head.norm.bias.data, None, out_dtype=F32,
                              eps=head.norm.eps, out=self.e_head_n.view(1, -1))
            offs = [0]
            for w in head.split_dims:
                offs.append(offs[-1] + w)
            for dim in dims:
                tab = self.tables[dim]
                ops.dec_gemv(tab, self.e_head_n[offs[dim]:offs[dim + 1]], self.logits)
                ops.dec_argmax_write(self.logits, tab.shape[0], self.seq2d, dim, pos)
        elif isinstance(head, TupleTokenLMHead):
            for dim in dims:
                lin = head.heads[keys[dim]]
                ops.dec_gemv(lin.weight.data, self.h, self.logits, bias=lin.bias.data)
                ops.dec_argmax_write(self.logits, lin.weight.shape[0], self.seq2d, dim, pos)
        else:
            raise NotImplementedError("decode engine: lm / lm-tied heads")
        ops.dec_add_pos(pos, 1)

    # -- the same step in ~40 fused launches -----------------------------------------------------------------------
    def _norm_args(self, norm):
        """(norm mode, gamma-or-gb-row, beta, eps) of a pre-norm for the fused GEMV prologue."""
        if self.ada and isinstance(norm, AdaptiveLayerNorm):
            return 2, self.gb_all[self.ada_rows[id(norm)]], None, norm.eps
        if isinstance(norm, AdaptiveLayerNorm):
            return 1, None, None, norm.eps          # no condition: plain normalisation (gamma = 1, beta = 0)
        return 1, norm.weight.data, norm.bias.data, norm.eps

    def _step_plan(self, dims: List[int]):
        """(fused_tail, latched, fold_cat, chains) of a fused step; builds the persistent launch's argument records on first use (no launch)."""
        m = self.m
        head, fn = m.lm_head, m.transformer.final_norm
        fused_tail = (isinstance(head, TupleTokenTiedLMHead) and head.reuse_projection and not isinstance(fn, nn.Identity)
                      and not self.reference_compat)
        # A dependent launch costs >= 4 us under graph replay whatever it does (profiles/, one-thread kernel), so three of them are
        # folded away: the position advance (latch protocol of ops.dec_step_begin; needs the fused head as the step's last launch), the
        # AdaLN-row GEMV (position-only input: rides in the first launch), the concatenation (prologue of the projection GEMV)
        latched = fused_tail and not self.legacy_launches
        fold_cat = isinstance(m.project_emb, nn.Linear) and not self.legacy_launches
        self.cur_dims = list(dims)
        chains = self._pair_chains(fold_cat and (latched or not self.ada_rows), fused_tail, latched and len(dims) <= 16) if self.pair_groups else {}
        return fused_tail, latched, fold_cat, chains

    def _one_launch_notes(self) -> bool:
        """True when `_steps` runs several notes as one launch (the whole note lives in the persistent launch)."""
        return bool(self.fused and self.multi_note and self.pair_groups and self.pair_embed and self.pair_head)

    def _steps(self, dims: List[int], U: int):
        """U consecutive notes (all of them to be decoded).  When a note is ONE persistent launch (embed .. head phases in it) the U notes are
        one launch too -- csrc/decode_layer.hip loops over them, the chosen tokens go from the head's winners to the next note's embed phase
        as granules -- else U steps."""
        if U > 1 and self.fused and self.multi_note and self.pair_groups:
            chains = self._step_plan(dims)[3]
            if self.pair_embed and self.pair_head and U <= 64:
                chains[0].launch(U)
                return
        step = self._step_fused if self.fused else self._step
        for _ in range(U):
            step(dims)

    def _step_fused(self, dims: List[int]):
        m, d, pos = self.m, self.dim, self.pos
        te, tr = m.token_emb, m.transformer
        has_norm = isinstance(te.norm, nn.LayerNorm)
        gam, bet, eps = (te.norm.weight.data, te.norm.bias.data, te.norm.eps) if has_norm else (None, None, 1e-5)
        head, fn = m.lm_head, tr.final_norm
        fused_tail, latched, fold_cat, chains = self._step_plan(dims)
        if self.pair_embed:   # embeddings, layer stack, LM head and position advance: ONE launch (csrc/decode_layer.hip)
            chains[0].launch()
            return
        en = m.emb_norm if isinstance(m.emb_norm, nn.LayerNorm) else None
        cat_args = dict(gamma=en.weight.data if en is not None else None, beta=en.bias.data if en is not None else None,
                        eps=en.eps if en is not None else 1e-5, ctx=self.ctx2d if m.context_emb_mode == "cat" else None,
                        style=self.style2d if m.style_emb_mode == "cat" else None)
        if self.legacy_launches:
            for si, (toks, off) in enumerate(((self.seq2d, 0), (self.masked2d, 1))):
                ops.dec_embed(self.tables, toks, pos, self.e_cat, row_off=off, gamma=gam, beta=bet, eps=eps)
                ops.dec_gemv(te.project_emb.weight.data, self.e_cat, self.proj_cat[si * d:(si + 1) * d], bias=te.project_emb.bias.data)
        elif latched:   # + position latch, + every AdaLN (gamma | beta) row of the step as a rider GEMV
            rider = (self.ada_W, self.style2d, 1, self.ada_b, self.gb_all.view(-1)) if self.ada_rows else None
            ops.dec_step_begin(self.tables, self.seq2d, self.masked2d, self.pos_next, pos, te.project_emb.weight.data, te.project_emb.bias.data,
                               self.proj_cat, gamma=gam, beta=bet, eps=eps, rider=rider)
        else:   # both sequences' tuple embeddings and their projection: one launch (4 before)
            ops.dec_embed_proj(self.tables, self.seq2d, self.masked2d, pos, te.project_emb.weight.data, te.project_emb.bias.data, self.proj_cat,
                               gamma=gam, beta=bet, eps=eps)
        if not self.pair_front:   # (else: first phase of the persistent launch)
            ops.dec_fused_gemv(te.project_multiemb.weight.data, self.proj_cat, self.x0 if fold_cat else self.x, bias=te.project_multiemb.bias.data,
                               pos=pos, y2=self.tok_emb, y2_ld=d)
        if self.ada_rows and not latched:   # every AdaLN (gamma | beta) row of this step: one GEMV over the stacked condition projections
            ops.dec_gemv(self.ada_W, self.style2d, self.gb_all.view(-1), bias=self.ada_b, pos=pos, x_ld=self.style2d.stride(0), x_off=1)
        if self.pair_front:
            pass                  # second phase of the persistent launch
        elif fold_cat:    # (LN(x0) | context row | style row) built in the prologue of the projection (x0 -> x: not in place)
            ops.dec_cat_gemv(m.project_emb.weight.data, self.x0, d, self.x, pos, bias=m.project_emb.bias.data, y2=self.hid[0], y2_ld=d, **cat_args)
        else:
            ops.dec_cat(self.x, d, self.xcat, pos, **cat_args)
            if isinstance(m.project_emb, nn.Linear):
                ops.dec_fused_gemv(m.project_emb.weight.data, self.xcat, self.x, bias=m.project_emb.bias.data, pos=pos, y2=self.hid[0], y2_ld=d)
            else:
                ops.dec_copy_row(self.xcat, self.x, pos, d)
                ops.dec_copy_row(self.x, self.hid[0], pos, d, dst_ld=d)
        ai = ci = 0
        n_layers = len(tr.layers)
        skip = 0
        for li, (lt, (norms, block, _res)) in enumerate(zip(tr.layer_types, tr.layers)):
            if skip:
                skip -= 1
                continue
            mode, g_, b_, eps_ = self._norm_args(norms[0])
            if li in chains:      # a chain of ('a', 'f') pairs: ONE persistent launch (csrc/decode_layer.hip)
                chains[li].launch()
                ai += chains[li].n
                skip = 2 * chains[li].n - 1
                continue
            if lt == 'c':
                self._cross_layer(ci, norms[0], block, True)
                ci += 1
            elif lt == 'a':
                wqkv = block._fused("_w_qkv", (block.to_q.weight, block.to_k.weight, block.to_v.weight)).data
                ops.dec_fused_gemv(wqkv, self.x, self.qkv, norm=mode, gamma=g_, beta=b_, eps=eps_)
                slopes = self._slopes(block)
                ops.dec_attn2(self.qkv, self.kc[ai], self.vc[ai], slopes, pos, self.o if self.legacy_launches else None, self.att_part,
                              self.att_counter, self.kmax2[ai], h=self.heads, kvh=self.kvh, scale=block.scale, splits=self.attn_splits)
                if self.legacy_launches:
                    ops.dec_fused_gemv(block.to_out.weight.data, self.o, self.x, residual=self.x)
                else:
                    ops.dec_attn_out(block.to_out.weight.data, self.att_part, self.x, h=self.heads, splits=self.attn_splits, residual=self.x)
                ai += 1
            else:
                lin = block.ff[0].proj if block.glu else block.ff[0][0]
                ops.dec_fused_gemv(lin.weight.data, self.x, self.g, norm=mode, gamma=g_, beta=b_, eps=eps_,
                                   bias=lin.bias.data if lin.bias is not None else None, glu=1 if block.glu else -1, act=block.act_code)
                if isinstance(block.ff[1], nn.LayerNorm):
                    ops.layernorm_fwd(self.g.view(1, -1), block.ff[1].weight.data, block.ff[1].bias.data, None, out_dtype=F32,
                                      eps=block.ff[1].eps, out=self.g.view(1, -1))
                out = block.ff[3]
                nxt_attn = li + 1 < n_layers
                ops.dec_fused_gemv(out.weight.data, self.g, self.x, bias=out.bias.data if out.bias is not None else None, residual=self.x,
                                   pos=pos, y2=self.hid[ai] if nxt_attn else None, y2_ld=d if nxt_attn else 0)
        if fused_tail:
            mode, g_, b_, eps_ = self._norm_args(fn)
            if not self.pair_tail:   # (else: last phase of the persistent launch)
                ops.dec_fused_gemv(self.head_Wt, self.x, self.e_head, norm=mode, gamma=g_, beta=b_, eps=eps_, pos=pos,
                                   xn_out=self.hid[-1], xn_ld=d)
            if self.pair_tail and self.pair_head:   # (the head was the last phase of the persistent launch, position advance included)
                return
            offs = [0]
            for w in head.split_dims:
                offs.append(offs[-1] + w)
            if self.sampling is not None:
                ops.dec_head_sample([self.tables[dim] for dim in dims], [offs[dim] for dim in dims], list(dims), te.total_emb_dim, self.e_head,
                                    head.norm.weight.data, head.norm.bias.data, head.norm.eps, self.seq2d, pos, self.head_part,
                                    self.head_counter, self.head_logits, self.sampling["topk"], self.seed_dev,
                                    temperature=self.sampling["temperature"], slabs=self.head_slabs,
                                    pos_next=self.pos_next if latched else None)
            else:
                ops.dec_head([self.tables[dim] for dim in dims], [offs[dim] for dim in dims], list(dims), te.total_emb_dim, self.e_head,
                             head.norm.weight.data, head.norm.bias.data, head.norm.eps, self.seq2d, pos, self.head_part, self.head_counter,
                             slabs=self.head_slabs, pos_next=self.pos_next if latched else None)
            if not latched:
                ops.dec_add_pos(pos, 1)
            return
        if self.sampling is not None:
            raise NotImplementedError("decode engine: sampling needs the fused tied LM head")
        # other head / norm combinations: the unfused tail
        if not isinstance(fn, nn.Identity):
            self._ln(self.x, fn, self.h, True)
        else:
            ops.dec_copy_row(self.x, self.h, pos, d)
        ops.dec_copy_row(self.h, self.hid[-1], pos, d, dst_ld=d)
        self._head_tail(dims)

    def _pair_chains(self, front_ok: bool = False, tail_ok: bool = False, head_ok: bool = False):
        """{index of the first layer of a chain of ('a', 'f') pairs: ops.DecPairChain}: consecutive pairs the persistent launch can take
        run as ONE launch.  Built on first use (the eager warm-up step), then reused (hipGraph capture included).  When one chain covers
        the whole decoder, the note's two input projections (front_ok) and the LM head's input projection (tail_ok) become phases of the
        same launch (spn_dec_chain_ext): self.pair_front / self.pair_tail."""
        if self.pair_chains:
            return self.pair_chains
        tr, d = self.m.transformer, self.dim
        types, n_layers = list(tr.layer_types), len(tr.layers)

        def eligible(li):
            if li + 1 >= n_layers or types[li] != 'a' or types[li + 1] != 'f':
                return False
            ff = tr.layers[li + 1][1]
            return bool(ff.glu) and not isinstance(ff.ff[1], nn.LayerNorm) and ff.act_code in (0, 1)

        n_pairs, li, ai_of = 0, 0, {}
        ai = 0
        for k, t_ in enumerate(types):
            if t_ == 'a':
                ai_of[k] = ai
                ai += 1
        while li < n_layers:
            if not eligible(li) or n_pairs >= 32:
                li += 1
                continue
            start, records = li, []
            while eligible(li) and n_pairs < 32:
                (norms, block, _), (fnorms, ff, _) = tr.layers[li], tr.layers[li + 1]
                mode, g_, b_, eps_ = self._norm_args(norms[0])
                m2, g2, b2_, e2 = self._norm_args(fnorms[0])
                lin, out = ff.ff[0].proj, ff.ff[3]
                wqkv = block._fused("_w_qkv", (block.to_q.weight, block.to_k.weight, block.to_v.weight)).data
                a_i = ai_of[li]
                nxt_attn = li + 2 < n_layers
                records.append(dict(
                    Wqkv=wqkv, ld_qkv=wqkv.stride(0), Wo=block.to_out.weight.data, ld_o=block.to_out.weight.stride(0),
                    W1=lin.weight.data, ld_1=lin.weight.stride(0), b1=lin.bias.data if lin.bias is not None else None,
                    W2=out.weight.data, ld_2=out.weight.stride(0), b2=out.bias.data if out.bias is not None else None,
                    slopes=self._slopes(block), kcache=self.kc[a_i], vcache=self.vc[a_i], kmax2=self.kmax2[a_i], jlo=self.pair_jlo[a_i] if os.environ.get("SPN_DEC_PAIR_JLO", "1") != "0" else None,
                    norm1=mode, gam1=g_, bet1=b_, eps1=eps_, norm2=m2, gam2=g2, bet2=b2_, eps2=e2, x=self.x,
                    y2=self.hid[a_i + 1] if nxt_attn else None, y2_ld=d if nxt_attn else 0, d=d, h=self.heads, kvh=self.kvh,
                    inner=self.g.numel(), S=self.attn_splits, act=ff.act_code, scale=block.scale, pos=self.pos, tick=self.pair_tick,
                    layer=n_pairs, bump=0, err=self.pair_err,
                    stamps=self.pair_stamps[a_i] if self.pair_stamps is not None else None, **self.pair_g))
                n_pairs += 1
                li += 2
            self.pair_chains[start] = records
        if self.pair_chains:
            last = max(self.pair_chains)
            self.pair_chains[last][-1]["bump"] = 1          # the last pair of the note advances the epoch counter
            ext = {}
            m, te = self.m, self.m.token_emb
            whole = len(self.pair_chains) == 1 and 0 in self.pair_chains and 2 * len(self.pair_chains[0]) == n_layers and n_pairs <= 30
            if whole and os.environ.get("SPN_DEC_PAIR_EXT", "1") != "0":
                cw = m.context_emb_dim if (m.context_emb_mode == "cat" and self.ctx2d is not None) else 0
                sw = m.style_emb_dim if (m.style_emb_mode == "cat" and self.style2d is not None) else 0
                Wm = te.project_multiemb.weight.data
                if front_ok and Wm.shape[1] <= 1024 and Wm.shape[1] % 4 == 0 and (d + cw + sw) <= 2048 and (d + cw + sw) % 4 == 0:
                    en = m.emb_norm if isinstance(m.emb_norm, nn.LayerNorm) else None
                    Wp = m.project_emb.weight.data
                    ext.update(Wm=Wm, ld_m=Wm.stride(0), bm=te.project_multiemb.bias.data, xin=self.proj_cat, Km=Wm.shape[1],
                               y2m=self.tok_emb, y2m_ld=d, Wp=Wp, ld_p=Wp.stride(0), bp=m.project_emb.bias.data,
                               cat_gamma=en.weight.data if en is not None else None, cat_beta=en.bias.data if en is not None else None,
                               cat_eps=en.eps if en is not None else 1e-5,
                               ctx=self.ctx2d if cw else None, ctx_ld=self.ctx2d.stride(0) if cw else 0, ctx_w=cw,
                               style=self.style2d if sw else None, style_ld=self.style2d.stride(0) if sw else 0, style_w=sw,
                               y2p=self.hid[0], y2p_ld=d, gf=self.pair_g2["gf"], gxf=self.pair_g2["gxf"])
                    self.pair_front = True
                if tail_ok and self.head_Wt.shape[0] <= 16 * self.heads * self.attn_splits:
                    mode, g_, b_, eps_ = self._norm_args(tr.final_norm)
                    ext.update(Wh=self.head_Wt, ld_h=self.head_Wt.stride(0), Nh=self.head_Wt.shape[0], normh=mode, gamh=g_, beth=b_, epsh=eps_,
                               e_out=self.e_head, xn_out=self.hid[-1], xn_ld=d)
                    self.pair_tail = True
                    # ... and the LM head itself -- arg-max, or (round 5) top-k sampling with the decision of spn_dec_head_sample inside the
                    # launch -- so that a note is TWO launches
                    if (head_ok and self._sampling_fits_the_launch() and te.total_emb_dim == self.head_Wt.shape[0] and te.total_emb_dim <= 2048
                            and te.total_emb_dim % 4 == 0 and os.environ.get("SPN_DEC_PAIR_HEAD", "1") != "0"):
                        ext.update(self._head_ext(), ge=self.pair_g2["ge"], gh=self.pair_g2["gh"])
                        if self.sampling is not None:
                            ext.update(gl=self.pair_g2["gl"], stopk=self.sampling["topk"], sinv_temp=1.0 / self.sampling["temperature"],
                                       sseed=self.seed_dev)
                        self.pair_head = True
                # ... and, in front, the two token-tuple embeddings with their projection (+ the NEXT note's AdaLN rows): ONE launch per note
                if self.pair_front and self.pair_head and os.environ.get("SPN_DEC_PAIR_EMBED", "1") != "0":
                    We = te.project_emb.weight.data
                    N, D = We.shape
                    nA = self.heads * self.attn_splits
                    R = -(-2 * N // (nA * 8))
                    K_r = self.ada_W.shape[1] if self.ada_rows else 4
                    if (R in (1, 2) and N % (8 * R) == 0 and 2 * N == Wm.shape[1] and D % 4 == 0 and D <= 2048 and We.stride(0) % 4 == 0
                            and K_r % 4 == 0 and K_r <= 256):
                        has_norm = isinstance(te.norm, nn.LayerNorm)
                        ext.update(self._embed_ext(), eD=D, eN=N, eR=R, egamma=te.norm.weight.data if has_norm else None,
                                   ebeta=te.norm.bias.data if has_norm else None, eeps=te.norm.eps if has_norm else 1e-5,
                                   We=We, ld_e=We.stride(0), be=te.project_emb.bias.data, gin=self.pair_g2["gin"], gt=self.pair_g2["gt"])
                        if self.ada_rows:
                            ext.update(rW=self.ada_W, r_ldw=self.ada_W.stride(0), rN=self.ada_W.shape[0], rK=K_r, rbias=self.ada_b,
                                       ry=self.gb_both, ada_par=self.gb_both.stride(0))
                        for rec in self.pair_chains[0]:          # no launch in front any more: the position comes from where the head leaves it
                            rec["pos"] = self.pos_next
                        self.pair_embed = True
            self.pair_chains = {k: ops.DecPairChain(v, self.dev, ext if (ext and k == 0) else None) for k, v in self.pair_chains.items()}
        return self.pair_chains

    def _sampling_fits_the_launch(self) -> bool:
        """Arg-max always; sampling when every decoded key's logits fit the head phase's hand-off (csrc/decode_layer.hip: vocabularies up to
        1024 ids, at most 64 rows per wave of a slab) and SPN_DEC_PAIR_SAMPLE is not 0."""
        if self.sampling is None:
            return True
        if os.environ.get("SPN_DEC_PAIR_SAMPLE", "1") == "0":
            return False
        dims = list(self.cur_dims)
        n_c = self.g.numel() // 32
        sl = min(16, max(1, n_c // max(1, len(dims))))
        for dim in dims:
            v = self.tables[dim].shape[0]
            um = -(-v // (8 * sl))
            if v > 1024 or 8 * um > 512 or 8 * sl * um > 1024:
                return False
        return True

    def _drop_chains(self):
        """Forget the argument records of the persistent launch (rebuilt by the next step): its set of phases is about to change."""
        if self.pair_groups:
            self.pair_chains, self.pair_front, self.pair_tail, self.pair_head, self.pair_embed = {}, False, False, False, False

    def _head_ext(self):
        """The head-phase fields of spn_dec_chain_ext for the CURRENT run (token buffer, tables and decoded keys change from run to run)."""
        te, head = self.m.token_emb, self.m.lm_head
        dims = list(self.cur_dims)
        offs = [0]
        for w in head.split_dims:
            offs.append(offs[-1] + w)
        tabs = [self.tables[dim] for dim in dims]
        return dict(hn=len(dims), hD=te.total_emb_dim, htable=tabs, hV=[t.shape[0] for t in tabs], hwidth=[t.shape[1] for t in tabs],
                    hcol0=[offs[dim] for dim in dims], hdim=dims, hgamma=head.norm.weight.data, hbeta=head.norm.bias.data, heps=head.norm.eps,
                    hban=0b11, tokens=self.seq2d, tok_ld=self.seq2d.stride(0), mask_id=1, pos_next=self.pos_next)

    def _embed_ext(self):
        """The per-run fields of the embed phase (token buffers, tables, style rows)."""
        if self.seq2d.stride(0) != self.masked2d.stride(0):
            raise ValueError("decode engine: the two token arrays must share their row stride")
        tabs = list(self.tables)
        col0 = [0]
        for t_ in tabs:
            col0.append(col0[-1] + t_.shape[1])
        kw = dict(en=len(tabs), etable=tabs, ewidth=[t_.shape[1] for t_ in tabs], ecol0=col0[:-1], tok_a=self.seq2d, tok_b=self.masked2d,
                  etok_ld=self.seq2d.stride(0))
        if self.ada_rows:
            kw.update(rx=self.style2d, rx_ld=self.style2d.stride(0), rx_rows=self.style2d.shape[0])
        return kw

    def _refresh_head_ext(self, dims):
        """Before the notes of a run: point the head / embed phases of an already built chain at this run's token buffers / tables / keys."""
        self.cur_dims = list(dims)
        if self.pair_groups and self.pair_chains and self.pair_head:
            kw = self._head_ext()
            if self.pair_embed:
                kw.update(self._embed_ext())
            self.pair_chains[0].update_ext(**kw)

    def _prime_note(self, dims, t0: int):
        """Before the first note (position t0) of a run.  One launch per note (pair_embed): every note's launch computes the NEXT note's
        AdaLN rows, so the first note's rows are computed here, into the row set of its parity."""
        self.cur_dims = list(dims)
        if not (self.fused and self.pair_groups):
            return
        if not self.pair_chains:
            self._step_plan(dims)                    # builds the chains (no launch)
        if self.pair_embed and self.ada_rows:
            ops.dec_gemv(self.ada_W, self.style2d, self.gb_both[t0 & 1].view(-1), bias=self.ada_b, pos=self.pos_next,
                         x_ld=self.style2d.stride(0), x_off=1)

    def _pair_failed(self) -> bool:
        """True when a hand-off poll of the persistent layer launch ran into its bound (`*err` != 0: that launch and every later one of
        the render returned early, so whatever they were to produce is garbage).  The engine then drops the persistent launch for the
        rest of its life -- error word cleared, chains and the captured graph released -- and the CALLER re-runs the notes through the five
        launches per pair, which give bit-identical results (tests/test_dec_pair_gpu.py).  Typical causes: another process or stream
        held CUs for longer than the poll bound (~0.3 s), or a CU-masked environment in which not all workgroups are resident."""
        if not self.pair_groups:
            return False
        code = int(self.pair_err.item())
        if code == 0:
            return False
        import warnings
        warnings.warn(f"decode engine: a hand-off of the persistent layer launch timed out (code {code}); falling back to the five "
                      f"launches per layer pair for this engine (same tokens, ~25 % slower per note)", RuntimeWarning, stacklevel=3)
        self.pair_err.zero_()
        self.use_pair, self.pair_groups = False, 0
        self.pair_chains, self.pair_front, self.pair_tail, self.pair_head, self.pair_embed = {}, False, False, False, False
        self.graph = self.graph_multi = None
        self.pair_fallbacks += 1
        return True

    # -- public ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def run(self, tokens: torch.Tensor, tokens_masked: torch.Tensor, context: Optional[torch.Tensor],
            style: Optional[torch.Tensor], mask_token_id: int = 1, context_mask: Optional[torch.Tensor] = None):
        """tokens / tokens_masked: [1, L, K] int64 on the GPU; returns (filled tokens, number of decoded positions).
        context: [1, L, d] rows concatenated per note (context_emb_mode 'cat') or [1, n_ctx, d] attended as a whole ('attention',
        with its key mask `context_mask` [1, n_ctx])."""
        out = self._run_once(tokens, tokens_masked, context, style, mask_token_id, context_mask)
        if self._pair_failed():      # (clears the error, switches the persistent launch off) -> the same window through the five launches
            out = self._run_once(tokens, tokens_masked, context, style, mask_token_id, context_mask)
        return out

    def _run_once(self, tokens, tokens_masked, context, style, mask_token_id, context_mask):
        m = self.m
        L = tokens.shape[1]
        self._alloc(L)
        if self.cross:
            if context is None:
                raise ValueError("decode engine: a cross-attending decoder needs its context")
            self._project_context(context[0], context_mask[0] if context_mask is not None else None)
        self.seq2d = tokens[0].clone().contiguous()
        self.masked2d = tokens_masked[0].contiguous()
        self.ctx2d = context[0].float().contiguous() if (context is not None and not self.cross) else None
        self.style2d = style[0].float().contiguous() if style is not None else None
        unmask = (self.seq2d == mask_token_id)
        rows = unmask.any(dim=1).nonzero().flatten()
        if rows.numel() == 0:
            self.n_steps = 0
            return self.seq2d[None], 0
        dims = unmask.any(dim=0).nonzero().flatten().tolist()
        last = int(rows.max())                      # one host read for the whole window
        with torch.no_grad():
            self.tables = [t.detach().float().contiguous() for t in build_tables(list(m.token_emb.embs.values()))]
        n_steps = last                              # positions t = 0 .. last-1 (predicting t+1)
        if self.reference_compat:
            first = int(rows.min())                 # the first position that holds MASK: the reference's first (cache-free) call
            tab = list(range(L))
            for p_ in range(first - 1, L):          # position p_ predicts idx = p_ + 1, decoded note number s = idx - first + 1
                s_, t_ = p_ + 2 - first, 1
                while t_ * (t_ + 1) // 2 < s_:
                    t_ += 1
                tab[p_] = first - 1 + (t_ - 1)
            self.stale_tab = torch.tensor(tab, device=self.dev, dtype=torch.int32)
        step = self._step_fused if self.fused else self._step
        self._refresh_head_ext(dims)
        self.pos2.zero_()
        if self.fused:
            self._prime_note(dims, 0)
        if self.use_graph and n_steps > 2:
            step(dims)                              # warm-up (also position 0), eager
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step(dims)                          # recorded, not executed; every replay reads *pos on the device
            # A graph launch costs ~7 us on the device between two replays, a dependency edge INSIDE a graph ~2 us: notes are replayed
            # GRAPH_NOTES at a time (the position lives on the device, so a graph of U steps is U copies of the same launches)
            U, rest = self.graph_notes, n_steps - 1
            if U > 1 and rest >= 2 * U:
                gU = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gU):
                    self._steps(dims, U)
                for _ in range(rest // U):
                    gU.replay()
                rest -= (rest // U) * U
                self.graph_multi = gU
            for _ in range(rest):
                g.replay()
            self.graph = g
        else:
            for _ in range(n_steps):
                step(dims)
        self.n_steps = n_steps
        return self.seq2d[None], n_steps

    def caches(self):
        """Caches in the reference's layout (TupleTransformerCaches fields) for the decoded prefix."""
        from .models.scoreperformer.transformer import TupleTransformerCaches
        n = self.n_steps
        def view(t, rows):   # reference layouts: [1, n, 64] multi-query, [1, h, n, 64] otherwise
            return t[None, :rows] if self.kvh == 1 else t[:rows].view(1, rows, self.kvh, 64).permute(0, 2, 1, 3)
        att, ai, ci = [], 0, 0
        for lt in self.m.transformer.layer_types:   # one entry per attention layer, in layer order ('c': the whole context)
            if lt == 'a':
                att.append(AttentionIntermediates(keys=view(self.kc[ai], n), values=view(self.vc[ai], n)))
                ai += 1
            elif lt == 'c':
                nctx = getattr(self, "ctx_rows", None) or self.xk[ci].shape[0]
                att.append(AttentionIntermediates(keys=view(self.xk[ci], nctx), values=view(self.xv[ci], nctx)))
                ci += 1
        return TupleTransformerCaches(token_emb=self.tok_emb[None, :n],
                                      transformer=TransformerIntermediates(hiddens=[h[None, :n] for h in self.hid], attention=att))


class RenderSession(GreedyDecoder):
    """The decode engine kept alive across the calls of the render loop.

    The reference's loop (`inference/generators.py:160-262`) calls `unmask_tokens` once per chord group with `torch.cat`-grown caches,
    re-slices them after every time window (`cut_caches`, 432-443) and rebuilds them from scratch whenever the context is cropped.
    Here the caches are static `[max_len, .]` buffers and the valid prefix is one integer: extending = writing the new rows and
    replaying the captured step, cutting = lowering `length`, cropping = `reset()` + recomputing the window with the same graph.
    """

    def __init__(self, decoder, max_len: int, dims: List[int], mask_token_id: int = 1, **kw):
        super().__init__(decoder, max_len, **kw)
        if self.reference_compat:
            raise NotImplementedError("render session: reference_compat exists for the single-call decode only (GreedyDecoder.run)")
        m, dev = self.m, self.dev
        self._alloc(max_len)
        if self.cross:
            # context_emb_mode 'attention' (inference/generators.py:230-240 with modules/transformer/transformer.py:92-93,201): the decoder
            # attends the score embeddings of the WINDOW, which grows by the new notes at every call -- projected keys / values live in
            # static [max_len, .] buffers, rows are appended as the window grows (a projection is row-wise), and the captured step reads the
            # number of valid rows from device memory (spn_dec_xattn_dyn).  ALiBi distances count from the end of the context, as the
            # reference's cache protocol gives them.
            n_cross = sum(1 for t in m.transformer.layer_types if t == 'c')
            self.xk = [torch.zeros(max_len, self.kvh * 64, device=dev, dtype=F32) for _ in range(n_cross)]
            self.xv = [torch.zeros(max_len, self.kvh * 64, device=dev, dtype=F32) for _ in range(n_cross)]
            self.nk_dev = torch.zeros(1, device=dev, dtype=torch.int32)
            self.ctx_rows = 0
        K = len(m.token_emb.embs)
        self.seq2d = torch.zeros(max_len, K, device=dev, dtype=torch.int64)
        self.masked2d = torch.zeros(max_len, K, device=dev, dtype=torch.int64)
        self.ctx2d = (torch.zeros(max_len, m.context_emb_dim, device=dev, dtype=F32)
                      if (getattr(m, "context_emb_dim", 0) and not self.cross) else None)
        self.style2d = torch.zeros(max_len, m.style_emb_dim, device=dev, dtype=F32) if getattr(m, "style_emb_dim", 0) else None
        with torch.no_grad():
            self.tables = [t.detach().float().contiguous() for t in build_tables(list(m.token_emb.embs.values()))]
        self.dims, self.mask_token_id = [int(d) for d in dims], mask_token_id
        self.length = 0            # positions 0 .. length-1 hold valid cache rows (token_emb, hiddens, keys, values)
        self.tag = None            # whatever identifies the window the cache belongs to (set by the caller)
        self._step_fn = self._step_fused if self.fused else self._step
        self.steps_run = 0
        self.prefilled_rows = 0
        self.prefill_min = 16      # shorter prefixes: the captured step is as fast

    def configure(self, sampling: Optional[dict]):
        """sampling=None: arg-max (greedy).  sampling=dict(k=int|None, thres=0.9, temperature=1.0, seed=int): `top_k` filtering
        (modules/sampling.py:28-33: k, or ceil((1 - thres) * V) per key) + one multinomial draw per token.  Switching between the
        two, or changing the temperature, re-records the step graph."""
        if sampling is None:
            if self.sampling is not None:
                self.sampling, self.graph, self.graph_multi = None, None, None
                self._drop_chains()              # the arg-max head may ride in the persistent launch again
            return
        head = self.m.lm_head
        if not (isinstance(head, TupleTokenTiedLMHead) and head.reuse_projection and self.fused):
            raise NotImplementedError("decode engine: sampling needs the fused tied LM head")
        import math
        V = [self.tables[d].shape[0] for d in self.dims]
        ks = [int(sampling["k"]) if sampling.get("k") is not None else math.ceil((1 - sampling.get("thres", 0.9)) * v) for v in V]
        ks = [max(1, min(k, v)) for k, v in zip(ks, V)]
        temperature = float(sampling.get("temperature", 1.0))
        if self.sampling is None or self.sampling["temperature"] != temperature:
            self.graph = self.graph_multi = None
            self._drop_chains()                  # the head phase of the persistent launch changes (arg-max <-> sampling, or its temperature)
            self.sampling = {"topk": torch.tensor(ks, device=self.dev, dtype=torch.int32), "temperature": temperature, "ks": ks,
                             "calls": int(sampling.get("seed", 0))}
        elif self.sampling["ks"] != ks:
            self.sampling["topk"].copy_(torch.tensor(ks, dtype=torch.int32))
            self.sampling["ks"] = ks

    def reset(self):
        self.length, self.tag = 0, None
        self.ctx_rows = 0
        for k in self.kmax2:
            k.zero_()

    def _extend_context(self, context: torch.Tensor, n: int):
        """Cross-attending decoders: keys / values of context rows ctx_rows .. n-1 (the rows of the new notes), then *nk_dev = n."""
        if n > self.ctx_rows:
            rows = context[self.ctx_rows:n].float().contiguous()
            ci = 0
            for lt, (_n, block, _r) in zip(self.m.transformer.layer_types, self.m.transformer.layers):
                if lt == 'c':
                    ops.gemm_f32(rows, block.to_k.weight.data, out=self.xk[ci][self.ctx_rows:n])
                    ops.gemm_f32(rows, block.to_v.weight.data, out=self.xv[ci][self.ctx_rows:n])
                    ci += 1
            self.ctx_rows = n
        self.nk_dev.fill_(n)

    def truncate(self, length: int):
        self.length = max(0, min(self.length, int(length)))

    @torch.no_grad()
    def load_caches(self, caches):
        """Adopt caches computed elsewhere (one batched module forward over a window: `TupleTransformerCaches`) as rows 0 .. n-1."""
        n = caches.token_emb.shape[1]
        if n > self.max_len:
            raise ValueError("caches longer than the session")
        if self.cross:
            raise NotImplementedError("render session: module caches of a cross-attending decoder are not adopted (re-prime note by note)")
        self.reset()
        self.tok_emb[:n].copy_(caches.token_emb[0])
        for dst, h in zip(self.hid, caches.transformer.hiddens):
            dst[:n].copy_(h[0])
        for i, a in enumerate(caches.transformer.attention):
            k, v = (a.keys[0], a.values[0]) if self.kvh == 1 else \
                (a.keys[0].permute(1, 0, 2).reshape(n, -1), a.values[0].permute(1, 0, 2).reshape(n, -1))
            self.kc[i][:n].copy_(k)
            self.vc[i][:n].copy_(v)
            self.kmax2[i].copy_(self.kc[i][:n].view(n, self.kvh, 64).pow(2).sum(-1).amax(0))   # the reach bound of dec_attn2
        self.length = n

    @torch.no_grad()
    def prefill(self, n: int):
        """Cache rows 0 .. n-1 from the tokens / embeddings already written to rows 0 .. n (by `decode`'s upload): the same fp32
        arithmetic as `_step`, but every operator runs ONCE over all n positions (exact-fp32 GEMMs instead of n GEMVs, causal
        attention with the position in the grid) -- what the reference does when it recomputes a cropped window in one forward.

        The pass is ~150 launches whose enqueueing (2-3 ms of host time) takes longer than their execution (~1.2 ms): under
        `use_graph` it is captured ONCE PER ROW COUNT and replayed afterwards -- every operand is a static buffer of the session, the
        intermediates live in the graphs' shared memory pool, and a full context window re-primes with a handful of distinct row counts
        (SPN_DEC_PREFILL_GRAPHS = 0: always eager; at most 64 counts are kept)."""
        graphs = getattr(self, "_prefill_graphs", None)
        if graphs is None:
            graphs = self._prefill_graphs = {}
            self._prefill_pool = None
        if not self.use_graph or os.environ.get("SPN_DEC_PREFILL_GRAPHS", "1") == "0":
            self._prefill_eager(n)
        elif n in graphs:
            graphs[n].replay()
        else:
            self._prefill_eager(n)                       # does the work this time (and warms every lazily built operand)
            if len(graphs) < 64:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                if self._prefill_pool is None:
                    self._prefill_pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(g, pool=self._prefill_pool):
                    self._prefill_eager(n)               # recorded, not executed
                graphs[n] = g
        self.length = n
        self.prefilled_rows += n

    def _prefill_eager(self, n: int):
        m, d = self.m, self.dim
        te, tr = m.token_emb, m.transformer
        cond = self.style2d[1:n + 1] if self.style2d is not None else None

        def ln(x, norm):
            if self.ada and isinstance(norm, AdaptiveLayerNorm):
                gb = ops.gemm_f32(cond, norm.linear.weight.data, bias=norm.linear.bias.data)
                return ops.layernorm_fwd(x, None, None, gb, out_dtype=F32, eps=norm.eps)[0]
            if isinstance(norm, AdaptiveLayerNorm):
                return ops.layernorm_fwd(x, None, None, None, out_dtype=F32, eps=norm.eps)[0]
            return ops.layernorm_fwd(x, norm.weight.data, norm.bias.data, None, out_dtype=F32, eps=norm.eps)[0]

        parts = []
        for toks in (self.seq2d[0:n], self.masked2d[1:n + 1]):
            e = torch.cat([t[toks[:, k]] for k, t in enumerate(self.tables)], dim=-1)          # gather + concat (data movement)
            if isinstance(te.norm, nn.LayerNorm):
                e = ops.layernorm_fwd(e, te.norm.weight.data, te.norm.bias.data, None, out_dtype=F32, eps=te.norm.eps)[0]
            parts.append(ops.gemm_f32(e, te.project_emb.weight.data, bias=te.project_emb.bias.data))
        x = ops.gemm_f32(torch.cat(parts, dim=-1), te.project_multiemb.weight.data, bias=te.project_multiemb.bias.data)
        self.tok_emb[:n].copy_(x)
        cols = [ln(x, m.emb_norm) if isinstance(m.emb_norm, nn.LayerNorm) else x]
        if m.context_emb_mode == "cat" and self.ctx2d is not None:
            cols.append(self.ctx2d[1:n + 1])
        if m.style_emb_mode == "cat" and self.style2d is not None:
            cols.append(self.style2d[1:n + 1])
        xcat = torch.cat(cols, dim=-1) if len(cols) > 1 else cols[0]
        if isinstance(m.project_emb, nn.Linear):
            x = ops.gemm_f32(xcat, m.project_emb.weight.data, bias=m.project_emb.bias.data)
        else:
            x = xcat[:, :d].contiguous()
        H, KV = self.heads * 64, self.kvh * 64
        ai = 0
        for lt, (norms, block, _res) in zip(tr.layer_types, tr.layers):
            if lt == 'a':
                self.hid[ai][:n].copy_(x)
                wqkv = block._fused("_w_qkv", (block.to_q.weight, block.to_k.weight, block.to_v.weight)).data
                qkv = ops.gemm_f32(ln(x, norms[0]), wqkv)
                self.kc[ai][:n].copy_(qkv[:, H:H + KV])
                self.vc[ai][:n].copy_(qkv[:, H + KV:H + 2 * KV])
                self.kmax2[ai].copy_(torch.maximum(self.kmax2[ai], self.kc[ai][:n].view(n, self.kvh, 64).pow(2).sum(-1).amax(0)))
                slopes = self._slopes(block)
                o = torch.empty(n, H, device=self.dev, dtype=F32)
                ops.dec_attn_rows(qkv, self.kc[ai], self.vc[ai], slopes, 0, o, h=self.heads, kvh=self.kvh, scale=block.scale)
                ops.gemm_f32(o, block.to_out.weight.data, out=x, accumulate=True)
                ai += 1
            else:
                lin = block.ff[0].proj if block.glu else block.ff[0][0]
                u = ops.gemm_f32(ln(x, norms[0]), lin.weight.data, bias=lin.bias.data if lin.bias is not None else None)
                inner = u.shape[1] // 2 if block.glu else u.shape[1]
                g = ops.dec_glu_rows(u, torch.empty(n, inner, device=self.dev, dtype=F32), inner, act=block.act_code, glu=block.glu)
                if isinstance(block.ff[1], nn.LayerNorm):
                    g = ops.layernorm_fwd(g, block.ff[1].weight.data, block.ff[1].bias.data, None, out_dtype=F32, eps=block.ff[1].eps)[0]
                out = block.ff[3]
                ops.gemm_f32(g, out.weight.data, bias=out.bias.data if out.bias is not None else None, out=x, accumulate=True)
        fn = tr.final_norm
        self.hid[-1][:n].copy_(ln(x, fn) if not isinstance(fn, nn.Identity) else x)

    @torch.no_grad()
    def decode(self, tokens: torch.Tensor, masked: torch.Tensor, context: Optional[torch.Tensor], style: Optional[torch.Tensor],
               n_new: int, batched_prefill: bool = True) -> torch.Tensor:
        """tokens / masked: int64 [Lin, K] (host or device), the last n_new rows carry MASK in the predicted dims; context / style:
        device rows aligned with them.  Positions < self.length are taken from the caches; a long uncached known prefix goes
        through `prefill` (batched) unless batched_prefill=False (note by note).  Returns the n_new filled rows (device)."""
        Lin, c = tokens.shape[0], self.length
        if Lin > self.max_len:
            raise ValueError(f"window of {Lin} notes exceeds the session's max_len {self.max_len}")
        if c > Lin - 1:
            raise ValueError("cache longer than the input window: truncate() or reset() first")
        self.seq2d[c:Lin].copy_(tokens[c:Lin], non_blocking=True)
        self.masked2d[c:Lin].copy_(masked[c:Lin], non_blocking=True)
        if self.cross:
            if context is None or context.shape[0] < Lin:
                raise ValueError("render session: a cross-attending decoder needs the context rows of the whole window")
            self._extend_context(context, Lin)
            batched_prefill = False   # the batched re-priming pass has no cross-attention: a cropped window is re-primed note by note
        if self.ctx2d is not None:
            self.ctx2d[c:Lin].copy_(context[c:Lin])
        if self.style2d is not None:
            self.style2d[c:Lin].copy_(style[c:Lin])
        if batched_prefill and c == 0 and Lin - 1 - n_new >= self.prefill_min:
            self.prefill(Lin - 1 - n_new)
            c = self.length
        if self.sampling is not None:   # a fresh stream per call: positions repeat after a cut, (seed, position, key) must not
            self.sampling["calls"] += 1
            self.seed_dev.fill_((self.sampling["calls"] * 0x9E3779B1) & 0x7FFFFFFF)
        steps = Lin - 1 - c
        for attempt in (0, 1):
            self.pos2.fill_(c)
            if self.fused and steps > 0:
                self._prime_note(self.dims, c)
            done = 0
            if self.use_graph and self.graph is None and steps > 0:
                self._step_fn(self.dims)                 # first step eager (warms every lazily built operand), then record once
                done = 1
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_fn(self.dims)
                self.graph = g
            rest = steps - done
            U = self.graph_notes
            if self.use_graph and U > 1 and rest >= U:        # GRAPH_NOTES notes per replay (see GreedyDecoder._run_once)
                if self.graph_multi is None:
                    if not done:                              # (a capture needs a current stream that has run the step before)
                        self.graph.replay()
                        rest -= 1
                    torch.cuda.synchronize()
                    gU = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gU):
                        self._steps(self.dims, U)
                    self.graph_multi = gU
                for _ in range(rest // U):
                    self.graph_multi.replay()
                rest -= (rest // U) * U
            if rest >= 2 and self.use_graph and self._one_launch_notes():
                # a chord group's few notes: ONE launch for all of them (spn_dec_pairs_notes, issued directly: the count differs from
                # call to call) instead of `rest` replays of the one-note graph
                self._steps(self.dims, rest)
                rest = 0
            for _ in range(rest):
                self.graph.replay() if self.use_graph else self._step_fn(self.dims)
            if attempt or not self._pair_failed():
                break
            # a hand-off of the persistent launch timed out somewhere in these steps: rows >= c of every cache, the decoded tokens and the
            # running key norms are garbage.  Rows < c came from earlier, checked calls (or the batched prefill, which has no persistent
            # launch): restore the inputs and the key-norm bound from them, then run the same steps through the five launches per pair.
            self.seq2d[c:Lin].copy_(tokens[c:Lin], non_blocking=True)
            for i, k in enumerate(self.kmax2):
                k.zero_() if c == 0 else k.copy_(self.kc[i][:c].view(c, self.kvh, 64).pow(2).sum(-1).amax(0))
        self.steps_run += steps
        self.length = Lin - 1
        self.n_steps = self.length
        return self.seq2d[Lin - n_new:Lin]
