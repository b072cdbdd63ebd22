// Data augmentation of the training corpus on the device: MIDITokenizer.augment (midi_tokenizer.py:364-417 for v1,
// :1023-1102 for v2), which MidiDataset.load_midi applies to every file it serves (train.py:62-63, `aug=True` by default),
// fused into the batch assembly (MidiDataset.__getitem__ slicing + collate_fn, train.py:69-90).  Integer work on token ids:
// the results are BIT-EXACT against the reference (tests/golden/augment_v{1,2}.npz come from the reference's own method).
//
// The reference augments a whole file and then cuts a window out of it.  Two of its rules look at the whole file:
//   (1) if ANY note's shifted pitch leaves 0..127 the file comes back unchanged (:1065-1066);
//   (2) a key signature on a track whose notes ALL sit on the drum channel gets sf = 0 (:1099-1104).
// Both depend only on per-file facts that no shift changes -- the lowest / highest pitch among the notes off channel 9, and
// the set of ORIGINAL channels per ORIGINAL track -- so they are computed ONCE per corpus (mh_augment_piece_stats: one
// workgroup per file, integer min / max / or in LDS) and every later batch is one launch that slices, augments, widens and
// pads B windows with row-local arithmetic (mh_augment_collate_windows).  HBM-bound, 2 bytes read + 8 written per token.
//
// `tab` (int32[MH_AUG_TAB]) carries what the kernels need of the tokenizer's tables (tokenizer.augment_table builds it from
// the public attributes, so either tokenizer version works); token columns are 1-based positions inside the octet, 0 = the
// event has no such parameter, event ids -1 = the tokenizer has no such event.
#include "common.h"

namespace {

enum {
  TB_T = 0,
  TB_EV = 1,        // 6 event ids: note, patch_change, control_change, set_tempo, time_signature, key_signature
  TB_TRACK_COL = 7,  // per event
  TB_CHAN_COL = 13,  // per event
  TB_NOTE_PITCH = 19,
  TB_NOTE_VEL = 20,
  TB_CC_CTRL = 21,
  TB_CC_VAL = 22,
  TB_TEMPO_BPM = 23,
  TB_KS_SF = 24,
  TB_KS_MI = 25,
  TB_TRACK0 = 26,
  TB_NTRACK = 27,
  TB_CHAN0 = 28,
  TB_NCHAN = 29,
  TB_PITCH0 = 30,
  TB_VEL0 = 31,
  TB_CTRL0 = 32,
  TB_VAL0 = 33,
  TB_BPM0 = 34,
  TB_NBPM = 35,
  TB_SF0 = 36,
  TB_MI0 = 37,
  TB_SIZE = 40,
};
constexpr int EV_NOTE = 0, EV_CC = 2, EV_TEMPO = 3, EV_KS = 5;
constexpr int MAX_TRACKS = 128;
constexpr int STATS = 2 + MAX_TRACKS;  // per file: min pitch, max pitch (notes off the drum channel), channel mask per track
constexpr int MAXT = 16;

__device__ inline int event_index(const int* tab, int id) {
#pragma unroll
  for (int e = 0; e < 6; ++e)
    if (tab[TB_EV + e] == id && id >= 0) return e;
  return -1;
}
// Python's % for a positive modulus
__device__ inline int pymod(int x, int m) {
  const int r = x % m;
  return r < 0 ? r + m : r;
}

__global__ __launch_bounds__(256) void augment_piece_stats_kernel(const int16_t* __restrict__ tokens,
                                                                  const int64_t* __restrict__ piece_off,
                                                                  const int* __restrict__ tabg, int* __restrict__ stats) {
  __shared__ int tab[TB_SIZE];
  __shared__ int s_min, s_max, s_mask[MAX_TRACKS];
  if (threadIdx.x < TB_SIZE) tab[threadIdx.x] = tabg[threadIdx.x];
  if (threadIdx.x < MAX_TRACKS) s_mask[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    s_min = 128;
    s_max = -1;
  }
  __syncthreads();
  const int T = tab[TB_T];
  const int64_t r0 = piece_off[blockIdx.x], r1 = piece_off[blockIdx.x + 1];
  const int tcol = tab[TB_TRACK_COL + EV_NOTE], ccol = tab[TB_CHAN_COL + EV_NOTE], pcol = tab[TB_NOTE_PITCH];
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
    const int16_t* row = tokens + r * T;
    if ((int)row[0] != tab[TB_EV + EV_NOTE]) continue;
    const int tr = (int)row[tcol] - tab[TB_TRACK0], c = (int)row[ccol] - tab[TB_CHAN0], p = (int)row[pcol] - tab[TB_PITCH0];
    if (c != 9) {
      atomicMin(&s_min, p);
      atomicMax(&s_max, p);
    }
    if (tr >= 0 && tr < MAX_TRACKS && c >= 0 && c < 32) atomicOr(&s_mask[tr], 1 << c);
  }
  __syncthreads();
  int* out = stats + (int64_t)blockIdx.x * STATS;
  if (threadIdx.x == 0) {
    out[0] = s_min;
    out[1] = s_max;
  }
  if (threadIdx.x < MAX_TRACKS) out[2 + threadIdx.x] = s_mask[threadIdx.x];
}

// one thread per output row (b, t)
__global__ __launch_bounds__(256) void augment_collate_kernel(const int16_t* __restrict__ tokens,
                                                              const int64_t* __restrict__ win_start,
                                                              const int64_t* __restrict__ win_len,
                                                              const int64_t* __restrict__ win_piece,
                                                              const int* __restrict__ shifts, const int* __restrict__ stats,
                                                              const int* __restrict__ tabg, int64_t* __restrict__ out, int64_t B,
                                                              int64_t L, int64_t pad_id) {
  __shared__ int tab[TB_SIZE];
  if (threadIdx.x < TB_SIZE) tab[threadIdx.x] = tabg[threadIdx.x];
  __syncthreads();
  const int T = tab[TB_T];
  const int64_t rows = B * L;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const int64_t b = r / L, t = r % L;
    int64_t* dst = out + r * T;
    if (t >= win_len[b]) {
      for (int j = 0; j < T; ++j) dst[j] = pad_id;
      continue;
    }
    int v[MAXT];
    const int16_t* src = tokens + (win_start[b] + t) * T;
    for (int j = 0; j < T; ++j) v[j] = (int)src[j];
    const int* sh = shifts + b * 6;  // pitch, velocity, cc value, bpm, track, channel (the reference's draw order)
    const int* st = stats + win_piece[b] * STATS;
    const int pitch_s = sh[0];
    // rule (1): a note off the drum channel whose shifted pitch leaves 0..127 anywhere in the FILE -> file unchanged
    const bool unchanged = st[1] >= 0 && (st[0] + pitch_s < 0 || st[1] + pitch_s > 127);
    const int e = unchanged ? -1 : event_index(tab, v[0]);
    if (e >= 0) {
      const int tcol = tab[TB_TRACK_COL + e], ccol = tab[TB_CHAN_COL + e];
      const int ntr = tab[TB_NTRACK], nch = tab[TB_NCHAN];
      int c0 = -1;
      int tr_new = -1;
      if (tcol) {
        tr_new = pymod(v[tcol] - tab[TB_TRACK0] + sh[4], ntr);
      }
      if (ccol) {
        c0 = v[ccol] - tab[TB_CHAN0];
        int c = pymod(c0 + sh[5], nch);
        if (c0 == 9) c = 9;
        else if (c == 9) c = pymod(9 + sh[5], nch);
        v[ccol] = tab[TB_CHAN0] + c;
      }
      if (e == EV_NOTE) {
        const int pc = tab[TB_NOTE_PITCH], vc = tab[TB_NOTE_VEL];
        int p = v[pc] - tab[TB_PITCH0];
        if (c0 != 9) p += pitch_s;
        int vel = v[vc] - tab[TB_VEL0] + sh[1];
        vel = max(1, min(127, vel));
        v[pc] = tab[TB_PITCH0] + p;
        v[vc] = tab[TB_VEL0] + vel;
      } else if (e == EV_CC) {
        const int cc = v[tab[TB_CC_CTRL]] - tab[TB_CTRL0];
        int val = v[tab[TB_CC_VAL]] - tab[TB_VAL0];
        if (cc == 1 || cc == 2 || cc == 7 || cc == 11) val = max(1, min(127, val + sh[2]));
        v[tab[TB_CC_VAL]] = tab[TB_VAL0] + val;
      } else if (e == EV_TEMPO) {
        int bpm = v[tab[TB_TEMPO_BPM]] - tab[TB_BPM0] + sh[3];
        bpm = max(1, min(tab[TB_NBPM] - 1, bpm));
        v[tab[TB_TEMPO_BPM]] = tab[TB_BPM0] + bpm;
      } else if (e == EV_KS) {
        int sf = v[tab[TB_KS_SF]] - tab[TB_SF0] - 7;
        const int mi = v[tab[TB_KS_MI]] - tab[TB_MI0];
        const int k = pymod(pymod(sf * 7, 12) + pitch_s, 12);  // sf2key, transposed
        sf = (k * 7) % 12;                                     // key2sf
        if (sf > 6 || (mi == 1 && sf >= 5)) sf -= 12;
        sf += 7;
        // rule (2): the SHIFTED track looked up among the ORIGINAL note tracks; all of that track's notes on channel 9 -> sf = 0
        if (tr_new >= 0 && tr_new < MAX_TRACKS && st[2 + tr_new] == (1 << 9)) sf = 7;
        v[tab[TB_KS_SF]] = tab[TB_SF0] + sf;
      }
      if (tcol) v[tcol] = tab[TB_TRACK0] + tr_new;
    }
    for (int j = 0; j < T; ++j) dst[j] = (int64_t)v[j];
  }
}

}  // namespace

extern "C" int mh_augment_piece_stats(const int16_t* tokens, const int64_t* piece_off, int64_t n_pieces, const int32_t* tab,
                                      int32_t* stats, void* stream) {
  MH_REQUIRE(n_pieces > 0 && n_pieces < (1ll << 31), "augment_piece_stats: n_pieces=%ld", (long)n_pieces);
  MH_REQUIRE(tokens && piece_off && tab && stats, "augment_piece_stats: null argument");
  augment_piece_stats_kernel<<<(unsigned)n_pieces, 256, 0, (hipStream_t)stream>>>(tokens, piece_off, tab, stats);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_augment_collate_windows(const int16_t* tokens, int64_t n_events, const int64_t* win_start,
                                          const int64_t* win_len, const int64_t* win_piece, const int32_t* shifts,
                                          const int32_t* stats, const int32_t* tab, int64_t* out, int64_t B, int64_t L, int T,
                                          int64_t pad_id, void* stream) {
  MH_REQUIRE(B > 0 && L > 0 && T > 1 && T <= MAXT && n_events > 0, "augment_collate_windows: bad shape B=%ld L=%ld T=%d", (long)B,
             (long)L, T);
  MH_REQUIRE(tokens && win_start && win_len && win_piece && shifts && stats && tab && out, "augment_collate_windows: null argument");
  const int64_t rows = B * L;
  const unsigned grid = (unsigned)((rows + 255) / 256 < 16384 ? (rows + 255) / 256 : 16384);
  augment_collate_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(tokens, win_start, win_len, win_piece, shifts, stats, tab, out, B, L,
                                                                 pad_id);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
