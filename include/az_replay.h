/*
 * az_replay.h — C ABI of the device-resident replay store.
 *
 * Replaces, next to the self-play engine, the consumer side of its output in the reference's Trainer:
 *   - the FIFO buffer of games                       train.py:226-236,295-298
 *   - Trainer.remove_duplicates                      train.py:156-201
 *   - the batch sampling of Trainer.net_step         train.py:107-113,119-120
 * so that finished games go engine -> replay -> training batch without leaving HBM and without being turned into
 * Python lists.  The network update itself (forward, loss, Adam: train.py:115-130) stays PyTorch.
 *
 * Semantics kept from the reference:
 *   - the buffer is a FIFO of GAMES: appending beyond `max_games` drops the oldest games (train.py:233-236);
 *   - an example's key is its action history (`state.information_state()`, game_utils.py:169), here a 64-bit
 *     hash of (length, actions) — duplicates are examples with equal history;
 *   - remove_duplicates walks the flattened buffer in order; for every key the pi vectors and the z values of
 *     its occurrences are summed IN THAT ORDER (float64, one rounding per addition) and divided by the count,
 *     the unique list is in first-occurrence order, AND the averaged pi / z are written back into the first
 *     occurrence stored in the buffer (the reference keeps a reference to that list object, train.py:191-197 —
 *     so the next generation's dedupe sees the averaged record).  All of it reproduced, bit for bit.
 *
 * Conventions as in az_engine.h (int status, az_replay_last_error, stream = hipStream_t as void*).
 */
#ifndef AZ_REPLAY_H
#define AZ_REPLAY_H

#include <stdint.h>

#include "az_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct az_replay az_replay;

typedef struct az_replay_config {
    int32_t struct_size;
    int32_t game, rows, cols; /* as az_config */
    int32_t device;
    int32_t reserved;
    int64_t max_games;        /* FIFO capacity in games (Trainer.n_games_buffer_max, train.py:38) */
    int64_t max_examples;     /* capacity in examples (plies) */
} az_replay_config;

typedef struct az_replay_stats {
    int64_t n_games, n_examples; /* currently stored */
    int64_t n_unique;            /* after the last az_replay_dedupe (0 before) */
    int64_t games_dropped;       /* total FIFO evictions */
    int64_t fault_flags;         /* AZ_REPLAY_FAULT_* raised on the device since the last az_replay_stats_get (reading clears them) */
} az_replay_stats;

#define AZ_REPLAY_FAULT_KEY_COLLISION 1u /* remove_duplicates met two different histories with one 64-bit key */
#define AZ_REPLAY_FAULT_BAD_INDEX 2u     /* az_replay_sample was given an index outside [0, n_unique) */

int az_replay_create(const az_replay_config *cfg, az_replay **out);
int az_replay_destroy(az_replay *r);
const char *az_replay_last_error(const az_replay *r);

/* Trainer.update_buffer_size / the FIFO trim (train.py:230-236,295-298): set the current capacity in games
 * (<= max_games); older games beyond it are dropped at the next append. */
int az_replay_set_capacity(az_replay *r, int64_t n_games);

/* Append every finished game of the engine's current generation, device to device, in game-id order
 * (`for examples in games: self.buffer.append(examples)`, train.py:226-227).  pi of an example is formed from the
 * recorded root visit counts with the reference's arithmetic (mcts.py:161-162, alphazerobot.py:13-14).
 * Synchronises `stream`. */
int az_replay_append_engine(az_replay *r, az_engine *e, void *stream);

/* Append games given as host arrays in the layout of az_example_view (used by tests and by the multi-rank gather). */
int az_replay_append_host(az_replay *r, const az_example_view *v, int32_t start_ply, void *stream);

/* Append n_games games from a DEVICE buffer in the packed layout of az_engine_export_device (this rank's own export,
 * or one rank's section of the all-gathered buffer): engine -> RCCL all-gather -> replay store without touching the
 * host.  The board geometry (max_plies, max_children) is the store's.  Synchronises `stream`. */
int az_replay_append_device(az_replay *r, const void *dev_buf, int64_t n_games, int32_t start_ply, void *stream);

/* Trainer.remove_duplicates over the whole (flattened) buffer.  Synchronises `stream`.  Records are grouped by the
 * 64-bit history hash; every member of a group is then checked against the group's first record (a second, independent
 * 64-bit hash of the history, the ply and the position): a mismatch - two different histories under one key, which the
 * reference, keying on the exact information-state string (train.py:177), would keep apart - returns AZ_E_DEVICE. */
int az_replay_dedupe(az_replay *r, void *stream);

/* The sampling of net_step (train.py:108-120): gather `batch` examples of the de-duplicated list into
 * x [batch][4][H][W] float32, pi [batch][A] float32, z [batch] float32 (all device).  `indices` (device int64
 * [batch], values in [0, n_unique)) are the `np.random.randint(len(flattened_buffer), size=batch_size)` draw;
 * pass NULL to draw them on the device (Philox, stream (seed, call counter)).  An index outside [0, n_unique) fills its
 * row with NaN and raises AZ_REPLAY_FAULT_BAD_INDEX (asynchronous: reported by az_replay_stats_get). */
int az_replay_sample(az_replay *r, const int64_t *indices, int32_t batch, uint64_t seed, float *x, float *pi,
                     float *z, void *stream);

/* Counters and device fault flags; synchronises the device.  Returns AZ_E_DEVICE when a fault flag is set; the flags are
 * reported ONCE (cleared by this call).  az_replay_dedupe clears AZ_REPLAY_FAULT_KEY_COLLISION when it starts, leaves a group
 * whose members differ untouched (nothing is averaged across different histories) and returns AZ_E_DEVICE for that pass. */
int az_replay_stats_get(az_replay *r, az_replay_stats *out);

/* Debug/parity read-back of the de-duplicated list (host arrays; any may be NULL): key hash, pi [n][A] float64,
 * z float64, index of the record in the flattened buffer, its bitboards and ply.  Returns n_unique or <0. */
int64_t az_replay_read_unique(az_replay *r, int64_t max_n, uint64_t *key, double *pi, double *z, int64_t *buffer_index,
                              uint64_t *bitboards, int32_t *ply);
/* Test hook for the collision guard of az_replay_dedupe: overwrite the 64-bit grouping key of stored example `index`
 * (its second hash, ply and position stay), so two different histories can be made to share a key. */
int az_replay_debug_set_key(az_replay *r, int64_t index, uint64_t key);
/* Read back one stored example of the flattened buffer (after write-back): pi [A], z. */
int az_replay_read_example(az_replay *r, int64_t index, double *pi, double *z);

#ifdef __cplusplus
}
#endif
#endif /* AZ_REPLAY_H */
