"""Input-side feature transforms on the device - counterpart of reference utilities/BoxTransforms.py (get_transforms,
:454-490) for the transforms the training drivers enable: ApplyLog, PadOrTrunc, TimeMask, FreqMask(fill "mean"), FreqShift,
ToTensor(unsqueeze_axis=0), Normalize(scaler).

The reference runs them per clip in NumPy inside the DataLoader workers; at > 10 k clips/s per GPU that cannot feed the step.
Here a whole batch is ONE kernel launch (sedt_box_transform: a workgroup per clip, the clip stays in LDS between the passes).
The RANDOM PARAMETERS are drawn on the host with np.random in exactly the order the reference's classes draw them
(TimeMask.randomize_parameters :380-383, FreqMask :410-413, FreqShift :437-443), so a seeded run augments every clip the same
way as the reference pipeline does."""
import numpy as np
import torch

from .. import lib as L

_AUG = np.dtype([('nframes_raw', np.int32), ('tm_t', np.int32), ('tm_t0', np.int32), ('fm_f', np.int32), ('fm_f0', np.int32),
                 ('fm_on', np.int32), ('fs_shift', np.int32), ('pad', np.int32)])


class DeviceBoxTransform(object):
    """frames: fixed number of frames (config.max_frames); scaler_mean / scaler_std: per-mel float64 vectors of the dataset
    Scaler (None: no normalisation); time_mask / freq_mask / freq_shift: enable the augmentations (their constructor
    defaults are the reference's: TimeMask(0.0, 0.1, p=0.2), FreqMask(0.03, 0.4, fill "mean", p=0.5), FreqShift(p=0.5,
    max_band=4, std=2)); apply_log=False takes inputs that are already in dB."""

    def __init__(self, frames, scaler_mean=None, scaler_std=None, time_mask=False, freq_mask=False, freq_shift=False,
                 apply_log=True, n_mels=64, device='cuda', tm=(0.0, 0.1, 0.2), fm=(0.03, 0.4, 0.5), fs=(0.5, 4, 0.0, 2.0)):
        self.frames, self.F, self.dev = frames, n_mels, torch.device(device)
        self.time_mask, self.freq_mask, self.freq_shift, self.apply_log = time_mask, freq_mask, freq_shift, apply_log
        self.tm, self.fm, self.fs = tm, fm, fs
        self.mean = self.std = None
        if scaler_mean is not None:
            self.mean = torch.as_tensor(np.asarray(scaler_mean, np.float64)).to(self.dev)
            self.std = torch.as_tensor(np.asarray(scaler_std, np.float64)).to(self.dev)

    def _upload(self, raw):
        """the per-clip parameter records through a ring of PINNED staging buffers: a copy from pageable memory blocks the host until
        everything queued before it has run - the host would lose its run-ahead over the GPU on every batch (measured on the C5
        step: 8.25 -> 8.65 ms)"""
        n = raw.size
        ring = self.__dict__.setdefault('_ring', {'pin': [], 'ev': [], 'dev': [], 'k': 0})
        if not ring['pin'] or ring['pin'][0].numel() < n:
            ring['pin'] = [torch.zeros(max(n, 4096), dtype=torch.uint8).pin_memory() for _ in range(4)]
            ring['dev'] = [torch.zeros(max(n, 4096), dtype=torch.uint8, device=self.dev) for _ in range(4)]
            ring['ev'] = [None] * 4
        k = ring['k']
        ring['k'] = (k + 1) % 4
        if ring['ev'][k] is not None:
            ring['ev'][k].synchronize()
        ring['pin'][k].numpy()[:n] = raw
        ring['dev'][k][:n].copy_(ring['pin'][k][:n], non_blocking=True)
        ring['ev'][k] = torch.cuda.Event()
        ring['ev'][k].record()
        return ring['dev'][k]

    def draw(self, nframes_raw):
        """one record of augmentation parameters for a clip, consuming np.random like the reference's transform objects"""
        r = np.zeros((), _AUG)
        r['nframes_raw'] = nframes_raw
        nf, nm = self.frames, self.F
        if self.time_mask:
            lo, hi, p = self.tm
            apply = np.random.uniform(0, 1) < p
            t = np.random.uniform(lo, hi)
            t0 = np.random.uniform(0, 1 - t)
            if apply:
                r['tm_t'], r['tm_t0'] = int(t * nf), int(t0 * nf)
        if self.freq_mask:
            lo, hi, p = self.fm
            apply = np.random.uniform(0, 1) < p
            f = np.random.uniform(lo, hi)
            f0 = np.random.uniform(0, 1 - f)
            if apply:
                r['fm_on'], r['fm_f'], r['fm_f0'] = 1, int(f * nm), int(f0 * nm)
        if self.freq_shift:
            p, max_band, mean, std = self.fs
            apply = np.random.uniform(0, 1) < p
            s = int(np.random.normal(mean, std))
            while abs(s) > max_band:
                s = int(np.random.normal(mean, std))
            if apply:
                r['fs_shift'] = s
        return r

    def draw_batch(self, nraws):
        """the records of a batch: the same np.random calls in the same order as draw() clip by clip, without a structured-array
        scalar per clip (64 clips: 1.4 ms -> 0.2 ms of host time per batch)"""
        rows = []
        u, nrm = np.random.uniform, np.random.normal
        nf, nm = self.frames, self.F
        for n in nraws:
            tm_t = tm_t0 = fm_f = fm_f0 = fm_on = fs = 0
            if self.time_mask:
                lo, hi, p = self.tm
                apply = u(0, 1) < p
                t = u(lo, hi)
                t0 = u(0, 1 - t)
                if apply:
                    tm_t, tm_t0 = int(t * nf), int(t0 * nf)
            if self.freq_mask:
                lo, hi, p = self.fm
                apply = u(0, 1) < p
                f = u(lo, hi)
                f0 = u(0, 1 - f)
                if apply:
                    fm_on, fm_f, fm_f0 = 1, int(f * nm), int(f0 * nm)
            if self.freq_shift:
                p, max_band, mean, std = self.fs
                apply = u(0, 1) < p
                s_ = int(nrm(mean, std))
                while abs(s_) > max_band:
                    s_ = int(nrm(mean, std))
                if apply:
                    fs = s_
            rows.append((n, tm_t, tm_t0, fm_f, fm_f0, fm_on, fs, 0))
        return np.asarray(rows, np.int32).view(_AUG).reshape(-1)

    def __call__(self, clips, params=None, out=None, staging=None):
        """clips: list of (T_raw, n_mels) float arrays / tensors (mel amplitudes), or a (B, T_raw, n_mels) tensor already on
        the device.  params: optional structured array of _AUG records (else drawn).  Returns (B, 1, frames, n_mels) f32."""
        if torch.is_tensor(clips) and clips.is_cuda:
            amp = clips.float().contiguous()
            B, stride = amp.shape[0], amp.shape[1]
            nraw = [stride] * B
        else:
            B = len(clips)
            nraw = [int(c.shape[0]) for c in clips]
            stride = max(nraw)
            host = staging if staging is not None else torch.zeros((B, stride, self.F), dtype=torch.float32).pin_memory()
            hv = host.numpy()
            for i, c in enumerate(clips):                       # plain memcpys into the pinned buffer (rows >= nraw[i] are never read)
                hv[i, :nraw[i]] = c.numpy() if torch.is_tensor(c) else c
            amp = host.to(self.dev, non_blocking=True)
        if params is None:
            params = self.draw_batch(nraw)
        params = np.ascontiguousarray(params)
        aug = self._upload(params.view(np.uint8).reshape(-1))
        if out is None:
            out = torch.empty((B, 1, self.frames, self.F), device=self.dev, dtype=torch.float32)
        L.check(L.load().sedt_box_transform(L.p(amp), stride, L.p(aug), L.p(self.mean), L.p(self.std), B, self.frames, self.F,
                                            int(self.apply_log), 1, 0.0, L.p(out), L.stream_ptr()), 'box_transform')
        return out


# ------------------------------------------------------------------------------------------------ SP-SEDT query patches
_JOB = np.dtype([('clip', np.int32), ('s_idx', np.int32), ('e_idx', np.int32), ('pad', np.int32)])


def random_patch_boxes(t, num_patches, mu=0.2, sigma=0.26, fixed_patch_size=False):
    """the (centre, length) patch boxes of one clip of ``t`` frames, drawn with np.random in the reference's order
    (data_utils/DataLoad.py:57-77, DataLoadDf.get_random_patch: 5 P normal lengths filtered to [0.05, 0.8), then one integer
    centre per kept length)"""
    if fixed_patch_size:
        l = np.asarray([128 / t] * num_patches)
    else:
        l = mu + sigma * np.random.randn(5 * num_patches)
        l = l[[0.05 <= i < 0.8 for i in l]][:num_patches]
    c = [np.random.randint(int(t * i / 2) + 1, int(t * (1 - i / 2))) / t for i in l]
    s, e = (c - l / 2) * t, (c + l / 2) * t
    s = [int(i) for i in s]
    e = [i + 128 for i in s] if fixed_patch_size else [int(i) for i in e]
    return [[(i + j) / (2 * t), (j - i) / t] for i, j in zip(s, e)]


class DeviceQuery(object):
    """reference utilities/BoxTransforms.py:315-360 (``Query``, the last transform of the SP-SEDT pipeline) for a whole batch in
    ONE launch: every box of every clip is cropped from the transformed clip, min-max normalised, quantised to 8 bits, resized
    to (128, F) with Pillow's bilinear arithmetic and de-normalised - bit-identical to the reference's PIL round trip
    (sedt_query_patches).  The row range of a box is computed on the host exactly as the reference does (float32 box values,
    ``int(s * t)``)."""

    def __init__(self, fixed_patch_size=False):
        self.fixed = bool(fixed_patch_size)

    def rows(self, box, t):
        c, l = np.float32(box[0]), np.float32(box[1])
        s, e = c - l / 2, c + l / 2
        s_idx, e_idx = int(s * t), int(e * t)
        if self.fixed:
            e_idx = min(t, s_idx + 128)
            s_idx = e_idx - 128
        elif s_idx >= e_idx:                                   # make sure the patch is not empty
            s_idx, e_idx = max(0, s_idx - 1), min(t, e_idx + 1)
        return s_idx, e_idx

    def __call__(self, data, boxes):
        """data (B, 1, T, F) f32 on the GPU (output of DeviceBoxTransform), boxes: per clip a (P, 2) array / tensor of (centre,
        length).  Returns patches (B, P, 1, 128, F) f32 - what the reference's collate stacks from ``label['patches']``."""
        if not data.is_cuda:
            raise RuntimeError('DeviceQuery runs on the MI355X HIP path only (no CPU fallback)')
        B, _, T, F = data.shape
        data = data.contiguous().float()
        P = len(boxes[0])
        bx = np.stack([(b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)).astype(np.float32) for b in boxes])
        assert bx.shape == (B, P, 2), 'every clip carries the same number of (centre, length) patch boxes'
        # the whole batch at once, in the float32 arithmetic of rows() (= the reference's box.numpy() scalars)
        c, l = bx[..., 0], bx[..., 1]
        half = l / np.float32(2)
        s_idx = ((c - half) * np.float32(T)).astype(np.int64)
        e_idx = ((c + half) * np.float32(T)).astype(np.int64)
        if self.fixed:
            e_idx = np.minimum(T, s_idx + 128)
            s_idx = e_idx - 128
        else:
            empty = s_idx >= e_idx                                      # make sure the patch is not empty
            s_idx = np.where(empty, np.maximum(0, s_idx - 1), s_idx)
            e_idx = np.where(empty, np.minimum(T, e_idx + 1), e_idx)
        bad = ~((0 <= s_idx) & (s_idx < e_idx) & (e_idx <= T)) | (self.fixed & (e_idx - s_idx != 128))
        if bad.any():
            b, k = np.argwhere(bad)[0]
            raise ValueError(f'patch box {bx[b, k].tolist()} gives rows [{s_idx[b, k]}, {e_idx[b, k]}) outside a clip of {T} frames')
        jobs = np.zeros((B * P,), _JOB)
        jobs['clip'] = np.repeat(np.arange(B, dtype=np.int32), P)
        jobs['s_idx'], jobs['e_idx'] = s_idx.reshape(-1), e_idx.reshape(-1)
        jd = torch.from_numpy(jobs.view(np.int32)).to(data.device, non_blocking=True)
        out = torch.empty((B, P, 1, 128, F), device=data.device, dtype=torch.float32)
        L.check(L.load().sedt_query_patches(data.data_ptr(), B, T, F, jd.data_ptr(), B * P, int(self.fixed), out.data_ptr(),
                                            L.stream_ptr()), 'query_patches')
        return out
