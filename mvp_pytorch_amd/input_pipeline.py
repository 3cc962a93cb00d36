"""Input pipeline of the pre-training step: the device-side counterpart of
`data_process` (oscar/run_pretrain_ml.py:474-513) and of the region-feature decode of
`OscarTSVDataset_C` (oscar/oscar_datasets_ml/oscar_tsv4.py:696-724, padding :332-352).

The reference decodes base64 -> float32 in DataLoader workers, collates 13 tensors per batch and
moves them with 13 `.to(device, non_blocking=True)` calls from pageable memory.  At MI355X step
rates (~13 k pairs/s => ~5.4 GB/s of f32 features per GPU) neither the Python base64 decode nor
pageable copies keep up, so here

  * the host only memcpy's the TSV rows' base64 TEXT into a pinned staging buffer (16-byte aligned
    per sample) and the twelve integer tensors into one pinned int64 block;
  * both go to the GPU with ONE asynchronous copy each on a dedicated copy stream, double buffered
    (`depth` slots), so batch i+1 is in flight while the step of batch i runs;
  * `mvptr_b64_decode_features` decodes, truncates to R rows, zero-pads and (optionally) writes the
    K-padded bf16 operand of the region-embedding GEMM, on the copy stream;
  * `get()` makes the compute stream wait on the slot's event and returns the batch dict that
    `train.model_inputs` takes.

Already-decoded float features (what the reference's own DataLoader yields) go through the same
pinned double buffer with `put_decoded`.
"""
import numpy as np
import torch

from . import hip

# the twelve integer tensors of a sample after img_feat, in __getitem__ order (oscar_tsv4.py:363-377)
INT_FIELDS = ("input_ids_a", "input_mask_a", "segment_ids_a", "lm_label_ids_a", "input_ids_b", "input_mask_b",
              "segment_ids_b", "lm_label_ids_b", "is_next", "is_img_match", "phrase_index", "image_index")


def _al16(n):
    return (n + 15) & ~15


class _Slot:
    pass


class PretrainBatchStager:
    """Double-buffered pinned staging + copy stream for batches of B samples.

    dims: dict(T, P, G, R) — text tokens, phrase slots, tag slots, regions; D = img_feature_dim.
    features: "f32" (img_feats f32 [B, R, D], what the reference hands the model), "bf16" (the
    K-padded bf16 [B*R, ld] operand only, batch["img_feats_bf16"]) or "both".
    """

    def __init__(self, device, B, dims, D, depth=2, features="f32", text_capacity=None):
        if not torch.cuda.is_available():
            raise RuntimeError("PretrainBatchStager needs a HIP device: the decode kernel has no CPU fallback")
        hip.load()
        self.device = torch.device(device)
        self.B, self.D = B, D
        self.R = dims["R"]
        La, G, R = dims["T"] + dims["P"], dims["G"], dims["R"]
        self.widths = dict(input_ids_a=La, input_mask_a=La, segment_ids_a=La, lm_label_ids_a=La, input_ids_b=G,
                           input_mask_b=G + R, segment_ids_b=G, lm_label_ids_b=G + R, is_next=1, is_img_match=1,
                           phrase_index=2, image_index=2)
        self.int_off, off = {}, 0
        for k in INT_FIELDS:
            self.int_off[k] = off
            off += self.widths[k]
        self.int_width = off + 3          # + text offset, n_chars, num_boxes of the sample
        self.features = features
        self.ld_bf16 = (D + 7) & ~7
        per_sample = _al16(((R * 2 * D * 4 + 2) // 3) * 4)   # room for 2R boxes per image before truncation
        self.text_capacity = int(text_capacity or B * per_sample)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = []
        for _ in range(depth):
            s = _Slot()
            s.text_h = torch.empty(self.text_capacity + 16, dtype=torch.uint8).pin_memory()
            s.text_np = s.text_h.numpy()
            s.ints_h = torch.empty(B, self.int_width, dtype=torch.int64).pin_memory()
            s.feat_h = None   # pinned f32 [B, R, D], allocated by the first put_decoded
            s.text_d = torch.empty(self.text_capacity + 16, dtype=torch.uint8, device=self.device)
            s.ints_d = torch.empty(B, self.int_width, dtype=torch.int64, device=self.device)
            s.nb_d = torch.empty(B, dtype=torch.int32, device=self.device)
            s.feat_d = torch.empty(B, R, D, dtype=torch.float32, device=self.device) if features in ("f32", "both") else None
            s.bf16_d = (torch.empty(B * R, self.ld_bf16, dtype=torch.bfloat16, device=self.device)
                        if features in ("bf16", "both") else None)
            s.err_d = torch.zeros(1, dtype=torch.int32, device=self.device)
            s.ready = torch.cuda.Event()
            s.consumed = None      # recorded by release(): the consuming step has been enqueued up to here
            s.in_use = False       # handed out by get(), not yet release()d
            s.staged = False
            s.text_bytes = 0
            self.slots.append(s)
        self._put, self._get, self._pending = 0, 0, None

    # ------------------------------------------------------------------ host side
    def _pack_ints(self, s, samples):
        ih = s.ints_h
        for j, k in enumerate(INT_FIELDS):
            o, w = self.int_off[k], self.widths[k]
            col = torch.stack([torch.as_tensor(smp[1 + j]).reshape(-1) for smp in samples])
            if col.shape[1] != w:
                raise ValueError("field %s: width %d, expected %d" % (k, col.shape[1], w))
            ih[:, o:o + w] = col

    def _slot_for_put(self):
        if self._put - self._get >= len(self.slots):
            raise RuntimeError("PretrainBatchStager: all %d slots are staged; get() one first" % len(self.slots))
        s = self.slots[self._put % len(self.slots)]
        if s.in_use:
            raise RuntimeError("PretrainBatchStager: call release() after enqueuing the step that consumed get()'s batch")
        if s.staged:
            s.ready.synchronize()      # the previous H2D copy out of this slot's pinned buffers has finished
        self._put += 1
        return s

    def _begin_device_work(self, s):
        """copy-stream work of a slot starts once the step that read its device tensors is done —
        NOT after everything queued on the compute stream (that would serialise copy and step)."""
        if s.consumed is not None:
            self.copy_stream.wait_event(s.consumed)

    def put(self, samples):
        """samples: B tuples ((b64_text: bytes, num_boxes: int), ids_a, mask_a, seg_a, lab_a, ids_b, mask_b,
        seg_b, lab_b, is_next, is_img_match, phrase_index, image_index) — a TSV row's feature columns
        (oscar_tsv4.py:716-719: arr[1], arr[-1]) followed by the integer tensors of __getitem__."""
        if len(samples) != self.B:
            raise ValueError("expected %d samples, got %d" % (self.B, len(samples)))
        s = self._slot_for_put()
        self._pack_ints(s, samples)
        meta = s.ints_h[:, self.int_width - 3:]
        pos = 0
        for i, smp in enumerate(samples):
            text, nb = smp[0]
            n = len(text)
            if pos + _al16(n) > self.text_capacity:
                raise ValueError("feature text of the batch exceeds text_capacity=%d bytes" % self.text_capacity)
            s.text_np[pos:pos + n] = np.frombuffer(text, dtype=np.uint8)
            meta[i, 0], meta[i, 1], meta[i, 2] = pos, n, int(nb)
            pos += _al16(n)
        s.text_bytes = pos
        with torch.cuda.stream(self.copy_stream):
            self._begin_device_work(s)
            s.text_d[:pos + 16].copy_(s.text_h[:pos + 16], non_blocking=True)
            s.ints_d.copy_(s.ints_h, non_blocking=True)
            s.nb_d.copy_(s.ints_d[:, self.int_width - 1])
            s.err_d.zero_()
            hip.b64_decode_features(s.text_d, s.ints_d[:, self.int_width - 3].contiguous(),
                                    s.ints_d[:, self.int_width - 2].contiguous(), s.nb_d, self.R, self.D,
                                    out_f32=s.feat_d, out_bf16=s.bf16_d, err=s.err_d)
            s.ready.record(self.copy_stream)
        s.decoded, s.staged = True, True

    def put_decoded(self, samples):
        """samples: B tuples (img_feat f32 [R, D], <12 integer tensors>) — exactly what
        OscarTSVDataset_C.__getitem__ returns (oscar_tsv4.py:363-377)."""
        if len(samples) != self.B:
            raise ValueError("expected %d samples, got %d" % (self.B, len(samples)))
        if self.features != "f32":
            raise ValueError("put_decoded stages f32 features: construct the stager with features='f32'")
        s = self._slot_for_put()
        if s.feat_h is None:
            s.feat_h = torch.empty(self.B, self.R, self.D, dtype=torch.float32).pin_memory()
        self._pack_ints(s, samples)
        torch.stack([smp[0] for smp in samples], out=s.feat_h)
        with torch.cuda.stream(self.copy_stream):
            self._begin_device_work(s)
            s.feat_d.copy_(s.feat_h, non_blocking=True)
            s.ints_d.copy_(s.ints_h, non_blocking=True)
            s.ready.record(self.copy_stream)
        s.decoded, s.staged = False, True

    def put_collated(self, batch):
        """batch: dict of COLLATED host tensors as `DataLoader(collate_fn=default, pin_memory=True)` yields them —
        img_feats f32 [B, R, D] and the integer fields [B, width] — the reference's own loader output
        (run_pretrain_ml.py:474-513 moves each with .to(device)).  The copies are issued straight from the given
        (pinned) tensors on the copy stream; with features "bf16" / "both" the K-padded bf16 operand of the
        region-embedding GEMM is produced there too (mvptr_cast_pack), off the step's critical path, and the model
        takes it instead of casting the f32 features itself."""
        s = self._slot_for_put()
        if s.feat_d is None:
            s.feat_d = torch.empty(self.B, self.R, self.D, dtype=torch.float32, device=self.device)
        with torch.cuda.stream(self.copy_stream):
            self._begin_device_work(s)
            s.feat_d.copy_(batch["img_feats"], non_blocking=True)
            for k in INT_FIELDS:
                if k in batch:
                    o, w = self.int_off[k], self.widths[k]
                    s.ints_d[:, o:o + w].copy_(batch[k].reshape(self.B, w), non_blocking=True)
            if s.bf16_d is not None:
                hip.cast_pack(s.feat_d.view(self.B * self.R, self.D), dst=s.bf16_d)
            s.ready.record(self.copy_stream)
        s.decoded, s.staged, s.have_bf16 = False, True, s.bf16_d is not None
        # the input-only counts of the step, taken from the host tensors while they are at hand (the model then needs no
        # read-back for them, synthetic.host_counts)
        s.host_counts = s.word_rows = None
        if all(k in batch for k in ("input_mask_a", "input_mask_b", "lm_label_ids_a", "lm_label_ids_b")):
            from .synthetic import host_counts, word_rows
            s.host_counts = host_counts(batch)
            s.word_rows = word_rows(batch)

    # ------------------------------------------------------------------ device side
    def get(self, check=False):
        """The oldest staged batch as the dict `train.model_inputs` takes.  The current stream waits
        for the slot's copies and decode; no host synchronisation unless check=True (reads the
        decode kernel's error flag)."""
        if self._get >= self._put:
            raise RuntimeError("PretrainBatchStager.get(): nothing staged")
        s = self.slots[self._get % len(self.slots)]
        self._get += 1
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(s.ready)
        if check and s.decoded:
            e = int(s.err_d.item())
            if e:
                raise RuntimeError("feature decode failed (flag %d): %s" % (
                    e, "text shorter than num_boxes x D floats" if e & 1 else "character outside the base64 alphabet"))
        batch = {}
        for k in INT_FIELDS:
            o, w = self.int_off[k], self.widths[k]
            v = s.ints_d[:, o:o + w]
            batch[k] = v.reshape(-1) if k in ("is_next", "is_img_match") else v
        if s.feat_d is not None:
            batch["img_feats"] = s.feat_d
        if s.bf16_d is not None and (s.decoded or getattr(s, "have_bf16", False)):
            batch["img_feats_bf16"] = s.bf16_d
        if getattr(s, "host_counts", None) is not None and not s.decoded:
            batch["host_counts"] = s.host_counts
            batch["word_rows"] = s.word_rows
        s.in_use = True
        self._pending = s
        return batch

    def release(self):
        """Call right after the step that consumed the last get() has been ENQUEUED: marks the point
        on the compute stream after which the slot's device tensors may be overwritten."""
        s = getattr(self, "_pending", None)
        if s is not None:
            s.consumed = torch.cuda.Event()
            s.consumed.record(torch.cuda.current_stream(self.device))
            s.in_use = False
            self._pending = None

    def batches(self, sample_batches, decoded=False):
        """Generator over device batches: while the consumer's step on batch i runs, batch i+1 is
        packed on the host, copied and decoded on the copy stream.

            for batch in stager.batches(loader):
                train.pretrain_step(model, batch, ...)
        """
        put = self.put_decoded if decoded else self.put
        it = iter(sample_batches)
        first = next(it, None)
        if first is None:
            return
        put(first)
        while True:
            batch = self.get()
            yield batch            # the consumer enqueues its step here
            self.release()
            nxt = next(it, None)
            if nxt is None:
                return
            put(nxt)


def encode_features_b64(feat):
    """float32 [num_boxes, D] -> the TSV column text the dataset stores (base64 of the raw bytes)."""
    import base64
    a = np.ascontiguousarray(feat, dtype=np.float32)
    return base64.b64encode(a.tobytes())
