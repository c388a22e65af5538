"""Device-resident end-to-end path: frames -> CNN embeddings -> fused scorer -> per-frame scores -> selection.

The reference does this in two stages with a .npy round-trip (scripts/preprocess.py:74-81 ->
data/dataset.py:45-52 -> scripts/evaluate.py:12-15); here the whole batch of videos stays in HBM.
Videos are independent (each is its own LSTM recurrence and its own B=1 attention call), so a
batch of videos is rows concatenated + row offsets, and a multi-GPU run shards the list of videos.
"""
import numpy as np
import torch

from . import ops
from .evaluation.metrics import select_frames


class _HostStager:
    """Double-buffered upload of passes of frames from pinned host memory: a copy stream fills buffer i % 2 while the
    compute stream works on the other one; events order the two streams, nothing blocks the host."""

    def __init__(self, host_frames, max_frames, device, pull_workgroups=0, cache=None):
        self.host = host_frames
        self.pull_workgroups = int(pull_workgroups)   # > 0: uploads by avs_pull_copy_u8 with that many workgroups
        # the copy stream and the two staging buffers live as long as the pipeline (`cache`: a dict it owns): creating a
        # stream and 3.4 GB of buffers per call showed up as 2 - 3 % of a 1.8 s step
        cache = cache if cache is not None else {}
        shape = (max_frames,) + tuple(host_frames.shape[1:])
        bufs = cache.get("bufs")
        if bufs is None or bufs[0].device != device or bufs[0].shape[1:] != shape[1:] or bufs[0].shape[0] < max_frames:
            bufs = [torch.empty(shape, dtype=torch.uint8, device=device) for _ in range(2)]
            cache["bufs"] = bufs
        self.bufs = bufs
        if cache.get("stream") is None or cache["stream"].device != device:
            cache["stream"] = torch.cuda.Stream(device=device)
        self.copy_stream = cache["stream"]
        # the staging buffers come from the COMPUTE stream's allocator pool and are reused from call to call: the first
        # upload into each buffer must be ordered behind everything the compute stream has been given so far (kernels of
        # the previous call that still read them; later uploads are ordered by the `freed` events)
        self.copy_stream.wait_stream(torch.cuda.current_stream(device))
        self.filled = [torch.cuda.Event(), torch.cuda.Event()]
        self.freed = [None, None]
        self.counts = [0, 0]
        self.base = 0     # pass i lives in buffer (i + base) % 2 (base = the buffer a prefetched first pass sits in)

    def adopt(self, k, event, count):
        """Pass 0 of this call was uploaded by the PREVIOUS call (embed(..., next_batch=)) into buffer k."""
        self.base = k
        self.filled[k] = event
        self.counts[k] = count

    def upload(self, i, one_pass, host=None):
        """host: another pinned tensor to read from (the next call's frames, prefetched into the buffer this call's pass i
        would take)."""
        _, where, contiguous = one_pass
        k = (i + self.base) % 2
        src_all = self.host if host is None else host
        with torch.cuda.stream(self.copy_stream):
            if self.freed[k] is not None:
                self.copy_stream.wait_event(self.freed[k])    # the pass that used this buffer has finished
            if contiguous:
                cnt = where[1] - where[0]
                if self.pull_workgroups > 0:
                    ops.pull_copy(src_all[where[0]:where[1]], self.bufs[k][:cnt], self.pull_workgroups)
                else:
                    self.bufs[k][:cnt].copy_(src_all[where[0]:where[1]], non_blocking=True)
            else:
                # an index set: runs of consecutive frames (whole videos minus their tails, or the tails themselves), one
                # transfer per run
                cnt = len(where)
                idx = np.asarray(where, dtype=np.int64)
                cuts = np.flatnonzero(np.diff(idx) != 1) + 1
                starts = np.concatenate([[0], cuts])
                ends = np.concatenate([cuts, [cnt]])
                for a, b in zip(starts.tolist(), ends.tolist()):
                    src = src_all[int(idx[a]):int(idx[a]) + (b - a)]
                    if self.pull_workgroups > 0:
                        ops.pull_copy(src, self.bufs[k][a:b], self.pull_workgroups)
                    else:
                        self.bufs[k][a:b].copy_(src, non_blocking=True)
            self.filled[k].record(self.copy_stream)
        self.counts[k] = cnt

    def ready(self, i):
        k = (i + self.base) % 2
        torch.cuda.current_stream().wait_event(self.filled[k])
        return self.bufs[k][:self.counts[k]]

    def release(self, i):
        k = (i + self.base) % 2
        self.freed[k] = torch.cuda.Event()
        self.freed[k].record(torch.cuda.current_stream())


class FrameScoringPipeline:
    def __init__(self, visual_extractor, scorer, use_inception=True, chunk_frames=256, frames_per_group=1, streams=1):
        """visual_extractor: features.extractors.VisualFeatureExtractor (on the device);
        scorer: models.av_model.AVBiLSTMModel (on the device, eval mode).
        frames_per_group: BatchNorm micro-batch size inside one video (reference: 4 per shot; the
        per-frame scoring mode embeds every frame as its own one-frame shot => 1)."""
        self.visual = visual_extractor
        self.scorer = scorer
        self.use_inception = use_inception
        self.chunk_frames = int(chunk_frames)
        self.frames_per_group = int(frames_per_group)
        # frames in pinned host memory: the first pass's upload has no computation to hide behind, so the first pass is
        # a short one (its upload is the only exposed copy of the step)
        self.host_lead_frames = 1024
        # the uploads run as a pull kernel of that many workgroups (ops.pull_copy) on the copy stream; 0 = torch's copy_
        # (the copy engine).  Measured (tools/h2d_study.py, 45 143 frames, 4 steps): 0.977 of the HBM-resident rate with the
        # pull kernel x16 and a 1024-frame lead pass, 0.955 - 0.960 with the copy engine
        self.host_pull_workgroups = 16
        self._stager_cache = {}
        self.check_exchange = True      # score(): verify that no clustered-BatchNorm exchange timed out (one counter read)
        # streams = 2: consecutive passes of the ResNet trunk run on two HIP streams, pass i + 1 starting when pass i
        # has launched its layers 1-2: the HBM-bound half of one pass then shares the chip with the matrix-core-bound
        # half (layers 3-4) of the other.  Passes are independent (disjoint frames, disjoint rows of the output).
        self.streams = int(streams)
        self._side = None
        if self.streams == 2:
            # kernels whose workgroups WAIT for each other (the clustered BatchNorm) rely on one queue's in-order dispatch: a
            # group's workgroups are dispatched together, complete groups always finish.  Two queues feeding the chip at once can
            # each hold partial groups that fill the CUs and wait for partners the other queue's partial groups keep out - the
            # bounded waits would end it and score() would raise.  Two-stream passes therefore keep to the unclustered forms
            visual_extractor._resnet_runner.bn_cluster = False

    def _group_offsets(self, video_offsets):
        """BatchNorm groups never straddle a video: per video, groups of frames_per_group (+ remainder)."""
        offs = [0]
        for a, b in zip(video_offsets[:-1], video_offsets[1:]):
            cur = a
            while cur < b:
                cur = min(cur + self.frames_per_group, b)
                offs.append(cur)
        return offs

    def _uniform_sets(self, video_offsets):
        """Frames partitioned into sets of EQUAL-sized BatchNorm groups (the fast kernels want one group size per
        pass): [(group size, frame indices | (start, end) when contiguous)].  With frames_per_group = g a video of
        L frames gives L // g full groups and, when g does not divide L, one last group of L % g frames
        (extractors.py:52-56: the last micro-batch of a shot is simply shorter); the full groups of all videos form
        one set, the tails of each size another."""
        gf = self.frames_per_group
        offs = np.asarray(video_offsets, dtype=np.int64)
        n = int(offs[-1])
        if n == 0:
            return []
        if gf == 1:
            return [(1, (0, n))]
        lens = offs[1:] - offs[:-1]
        rem = lens % gf
        if not rem.any():
            return [(gf, (0, n))]
        full_end = offs[:-1] + (lens - rem)
        sets = []
        full = np.concatenate([np.arange(a, e) for a, e in zip(offs[:-1], full_end) if e > a] or [np.zeros(0, np.int64)])
        if full.size:
            sets.append((gf, full))
        for r in range(1, gf):
            tails = [np.arange(e, b) for e, b, q in zip(full_end, offs[1:], rem) if q == r]
            if tails:
                sets.append((r, np.concatenate(tails)))
        return sets

    @staticmethod
    def _pass_frames(count, chunk_frames, gsz):
        """Frames per pass for `count` frames in groups of gsz: the fewest passes of at most chunk_frames frames, made
        equal in size, whole groups (45 143 frames at 24 576 -> 2 x 22 572, not 24 576 + 20 567)."""
        cap = max(gsz, chunk_frames // gsz * gsz)
        npass = max(1, -(-count // cap))
        return min(cap, max(gsz, -(-(-(-count // npass)) // gsz) * gsz))

    def _build_passes(self, video_offsets, lead_ok):
        """[(group size, frame slice | index array, contiguous)]: per uniform set, equal passes of at most chunk_frames frames
        (whole groups); lead_ok: a short first pass in front (host frames whose first upload has nothing to hide behind)."""
        passes = []
        for gsz, where in self._uniform_sets(video_offsets):
            contiguous = isinstance(where, tuple)
            lo, hi = where if contiguous else (0, len(where))
            lead = 0
            if lead_ok and not passes and self.host_lead_frames > 0:
                lead = max(gsz, self.host_lead_frames // gsz * gsz)
                if lead * 2 >= hi - lo:
                    lead = 0                       # too few frames for a lead pass to be worth a launch sequence
            if lead:
                passes.append((gsz, (lo, lo + lead) if contiguous else where[lo:lo + lead], contiguous))
                lo += lead
            per_pass = self._pass_frames(hi - lo, self.chunk_frames, gsz)
            for a in range(lo, hi, per_pass):
                b = min(a + per_pass, hi)
                passes.append((gsz, (a, b) if contiguous else where[a:b], contiguous))
        return passes

    def embed(self, frames_u8, video_offsets, next_batch=None):
        """uint8 [N,224,224,3] -> fp32 [N,4096] on the device (ResNet-50 | Inception-v3 halves).
        frames_u8 on the device: read in place.  frames_u8 in PINNED host memory: every pass's frames are uploaded
        on a copy stream into one of two staging buffers while the previous pass computes (PCIe-inclusive path).
        next_batch = (pinned frames, offsets) of the batch the NEXT call will be given (a stream of batches): its first
        pass is uploaded while this call's last pass computes, so that no upload of the next call is exposed and it needs no
        short lead pass.  The caller must leave that tensor alone until the next call has consumed it."""
        n = frames_u8.shape[0]
        host = not frames_u8.is_cuda
        if host and not frames_u8.is_pinned():
            raise ValueError("host frames must be in pinned memory (tensor.pin_memory()) for the overlapped upload")
        dev = self.visual._resnet_runner.trunk[0].weight.device if host else frames_u8.device
        visual = torch.zeros((n, 4096), dtype=torch.float32, device=dev)
        pref = self._stager_cache.get("prefetched") if host else None
        if pref is not None and pref["key"] != (frames_u8.data_ptr(), n, tuple(video_offsets)):
            pref = None                            # another batch than the one announced: its first pass is uploaded afresh
        self._stager_cache.pop("prefetched", None)
        passes = self._build_passes(video_offsets, lead_ok=host and pref is None)
        next_first = None
        if host and next_batch is not None and next_batch[0].shape[0] > 0:
            if next_batch[0].is_cuda or not next_batch[0].is_pinned():
                raise ValueError("next_batch frames must be in pinned host memory")
            next_first = self._build_passes([int(v) for v in next_batch[1]], lead_ok=False)[0]
        sizes = [(p[1][1] - p[1][0]) if p[2] else len(p[1]) for p in passes + ([next_first] if next_first else [])]
        stage = _HostStager(frames_u8, max(sizes), dev, self.host_pull_workgroups, self._stager_cache) if host and passes else None
        if stage is not None:
            if pref is not None and stage.bufs is pref["bufs"]:
                stage.adopt(pref["k"], pref["event"], pref["count"])
            else:
                if pref is not None:               # the staging buffers were re-allocated meanwhile: start over with a lead pass
                    passes = self._build_passes(video_offsets, lead_ok=True)
                stage.upload(0, passes[0])
        # two-stream overlap: device-resident contiguous passes of the ResNet-only path (the common case)
        overlap = (self.streams == 2 and stage is None and not self.use_inception and len(passes) > 1
                   and all(p[2] for p in passes))
        if overlap:
            if self._side is None:
                self._side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            side, main, mid_prev = self._side, torch.cuda.current_stream(dev), None
        for i, (gsz, where, contiguous) in enumerate(passes):
            idx = None
            if stage is not None:
                if i + 1 < len(passes):
                    stage.upload(i + 1, passes[i + 1])     # overlaps with this pass's kernels
                elif next_first is not None:
                    # the NEXT call's first pass, into the buffer this call's (non-existent) pass i + 1 would take
                    stage.upload(i + 1, next_first, host=next_batch[0])
                    k = (i + 1 + stage.base) % 2
                    self._stager_cache["prefetched"] = {
                        "key": (next_batch[0].data_ptr(), next_batch[0].shape[0], tuple(int(v) for v in next_batch[1])),
                        "k": k, "event": stage.filled[k], "count": stage.counts[k], "bufs": stage.bufs}
                chunk = stage.ready(i)
                cnt = chunk.shape[0]
                if contiguous:
                    out = visual[where[0]:where[1]]
                else:
                    idx = torch.from_numpy(where).to(dev)
            elif contiguous:
                chunk, out = frames_u8[where[0]:where[1]], visual[where[0]:where[1]]
                cnt = where[1] - where[0]
            else:
                idx = torch.from_numpy(where).to(dev)
                chunk = frames_u8.index_select(0, idx)     # memory plumbing: gather the pass's frames
                cnt = len(where)
            if idx is not None:
                out = torch.empty((cnt, 4096), dtype=torch.float32, device=dev)
                if not self.use_inception:
                    out[:, 2048:].zero_()
            groups = torch.arange(0, cnt + 1, gsz, dtype=torch.int64)
            if overlap:
                # pass i on stream i % 2, behind: everything queued before embed() (main), the same stream's previous
                # pass (stream order), and the MIDDLE of pass i - 1 (its layers 1-2 launched)
                st = side[i % 2]
                st.wait_stream(main)
                if mid_prev is not None:
                    st.wait_event(mid_prev)
                mid = torch.cuda.Event()
                with torch.cuda.stream(st):
                    self.visual._resnet_runner.forward(chunk, groups, out=out[:, :2048], mid_hook=lambda: mid.record(st))
                chunk.record_stream(st)
                mid_prev = mid
                continue
            self.visual._resnet_runner.forward(chunk, groups, out=out[:, :2048])
            if self.use_inception:
                big = ops.resize_bilinear(chunk, 299, 299)
                self.visual._inception_runner.forward(big, out=out[:, 2048:])
            if idx is not None:
                visual.index_copy_(0, idx, out)
            if stage is not None:
                stage.release(i)
        if overlap:
            for st in side:
                main.wait_stream(st)
        return visual

    @torch.no_grad()
    def score(self, frames_u8, video_offsets, audio_rows=None, next_batch=None):
        """Per-frame importance scores fp32 [N] for videos given as frame offsets [V+1].  next_batch: see embed()."""
        video_offsets = [int(v) for v in video_offsets]
        visual = self.embed(frames_u8, video_offsets, next_batch=next_batch)
        if audio_rows is None:
            # AudioFeatureExtractor.forward literally returns zeros(296)
            audio_rows = torch.zeros((visual.shape[0], self.scorer.audio_fc[0].in_features), dtype=torch.float32,
                                     device=visual.device)
        seq = torch.tensor(video_offsets, dtype=torch.int64, device=visual.device)
        scores = self.scorer.score_rows(visual, audio_rows, seq, attn_batch=1)
        if self.check_exchange:
            # the clustered BatchNorm launches of this call: a bounded wait that ran out means partner tiles were not
            # co-resident (never seen; the dispatch order is not a HIP guarantee) - fail loudly instead of returning
            # features normalised with incomplete statistics.  Reads one counter back (the caller syncs for the scores anyway)
            bad = ops.cluster_exchange_errors(visual.device)
            if bad:
                raise RuntimeError(f"clustered BatchNorm: {bad} wave(s) gave up waiting for a partner tile's statistics")
            bad = ops.lstm_split_errors(visual.device)      # the scorer's recurrences split over four CUs, likewise
            if bad:
                raise RuntimeError(f"split LSTM recurrence: {bad} workgroup(s) gave up waiting for a partner's step vector")
        return scores

    @staticmethod
    def select(scores, video_offsets):
        """Selection rule of scripts/evaluate.py:26 per video, on the host (bit-exact numpy)."""
        host = scores.detach().cpu().numpy()
        return [select_frames(host[a:b]) for a, b in zip(video_offsets[:-1], video_offsets[1:])]
