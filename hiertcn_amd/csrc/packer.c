/* Host-side batch packer of the XING path (the lane queues of data_loader.py:115-272 in the reference): plain C, no GPU.
 *
 * hiertcn_amd/data_loader.py cuts every user into sessions once (array passes) and keeps, per session, its first event row, its kept
 * length and its mask.  What is left per batch is queue bookkeeping and a gather -- at the reference's 32 lanes that is ~25 Python-level
 * heap operations and ~15 NumPy calls on 6,400-element arrays: 90 us per batch on the GPU box's host against a 0.22 ms device step (a third
 * of the evaluation loop).  This file does exactly the same bookkeeping in C:
 *   - B lane queues of session numbers; a user's sessions go to the currently SHORTEST lane, lowest index on ties (np.argmin,
 *     data_loader.py:203) -- a binary heap keyed (length, lane), so 4096-lane batches stay cheap;
 *   - users are enqueued, at least one per batch, until every lane holds MORE than S sessions (:229);
 *   - a batch pops S sessions per lane and is written as ids x / y [B][S][Lmax] int32 + mask [B][S] float (include/hiertcn.h's layout):
 *     y = the session's item ids, x = y shifted by one step (null id first), zero past a SLOT's own width (the widest session of the slot).
 * The user order (arange or a permutation per pass) stays in Python: pk_fill() returns when the walk is exhausted and is called again
 * with the next pass's order.  tests/test_data_loader.py holds the packer to the reference loader's captured batches.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t* v;
  int64_t head, tail, cap;
} lane_t;

typedef struct {
  int B, S;
  lane_t* lanes;
  int64_t* heap_len; /* binary min-heap over (len, lane): absolute queue lengths */
  int32_t* heap_lane;
  int32_t* heap_pos; /* lane -> position in the heap */
  int64_t* qlen;     /* lane -> sessions queued */
  const int64_t* s_begin;
  const int64_t* s_len;
  const double* s_mask;
  const int64_t* ids;
  const int64_t* u_first;
  const int64_t* u_nsess;
} packer_t;

static int heap_less(const packer_t* p, int a, int b) {
  if (p->heap_len[a] != p->heap_len[b]) return p->heap_len[a] < p->heap_len[b];
  return p->heap_lane[a] < p->heap_lane[b];
}
static void heap_swap(packer_t* p, int a, int b) {
  const int64_t tl = p->heap_len[a];
  const int32_t tn = p->heap_lane[a];
  p->heap_len[a] = p->heap_len[b]; p->heap_lane[a] = p->heap_lane[b];
  p->heap_len[b] = tl; p->heap_lane[b] = tn;
  p->heap_pos[p->heap_lane[a]] = a;
  p->heap_pos[p->heap_lane[b]] = b;
}
static void heap_down(packer_t* p, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < p->B && heap_less(p, l, m)) m = l;
    if (r < p->B && heap_less(p, r, m)) m = r;
    if (m == i) return;
    heap_swap(p, i, m);
    i = m;
  }
}

/* version of this file's interface: hiertcn_amd/data_loader.py refuses a library that reports another one (a stale build after a change here) */
#define PK_ABI_VERSION 2
int pk_abi_version(void) { return PK_ABI_VERSION; }

void* pk_create(int B, int S, const int64_t* s_begin, const int64_t* s_len, const double* s_mask, const int64_t* ids, const int64_t* u_first,
                const int64_t* u_nsess) {
  packer_t* p = (packer_t*)calloc(1, sizeof(packer_t));
  if (!p) return NULL;
  p->B = B; p->S = S;
  p->lanes = (lane_t*)calloc((size_t)B, sizeof(lane_t));
  p->heap_len = (int64_t*)calloc((size_t)B, sizeof(int64_t));
  p->heap_lane = (int32_t*)calloc((size_t)B, sizeof(int32_t));
  p->heap_pos = (int32_t*)calloc((size_t)B, sizeof(int32_t));
  p->qlen = (int64_t*)calloc((size_t)B, sizeof(int64_t));
  if (!p->lanes || !p->heap_len || !p->heap_lane || !p->heap_pos || !p->qlen) {   /* (free(NULL) is a no-op) */
    free(p->lanes); free(p->heap_len); free(p->heap_lane); free(p->heap_pos); free(p->qlen);
    free(p);
    return NULL;
  }
  for (int b = 0; b < B; b++) { p->heap_lane[b] = b; p->heap_pos[b] = b; }   /* all lengths 0: lane order = heap order */
  p->s_begin = s_begin; p->s_len = s_len; p->s_mask = s_mask; p->ids = ids; p->u_first = u_first; p->u_nsess = u_nsess;
  return p;
}

void pk_destroy(void* h) {
  packer_t* p = (packer_t*)h;
  if (!p) return;
  for (int b = 0; b < p->B; b++) free(p->lanes[b].v);
  free(p->lanes); free(p->heap_len); free(p->heap_lane); free(p->heap_pos); free(p->qlen);
  free(p);
}

static int lane_push(lane_t* q, int64_t first, int64_t n) {
  if (q->tail + n > q->cap) {
    const int64_t live = q->tail - q->head;
    if (q->head > 0 && live + n <= q->cap) {   /* slide the live part to the front */
      memmove(q->v, q->v + q->head, (size_t)live * sizeof(int64_t));
    } else {
      int64_t cap = q->cap ? q->cap : 64;
      while (cap < live + n) cap *= 2;
      int64_t* nv = (int64_t*)malloc((size_t)cap * sizeof(int64_t));
      if (!nv) return -1;
      if (live) memcpy(nv, q->v + q->head, (size_t)live * sizeof(int64_t));
      free(q->v);
      q->v = nv; q->cap = cap;
    }
    q->head = 0; q->tail = live;
  }
  for (int64_t i = 0; i < n; i++) q->v[q->tail + i] = first + i;
  q->tail += n;
  return 0;
}

/* Enqueue users walk[*pointer ..] -- at least one when `force_one` -- until every lane holds more than S sessions.
 * Returns 0: done; 1: the walk ran out first (the caller starts the next pass and calls again with force_one = 0); -1: out of memory. */
int pk_fill(void* h, const int64_t* walk, int64_t n_users, int64_t* pointer, int force_one) {
  packer_t* p = (packer_t*)h;
  int first = force_one;
  for (;;) {
    if (!first && p->heap_len[0] > p->S) return 0;
    if (*pointer >= n_users) return 1;
    first = 0;
    const int64_t u = walk[*pointer];
    const int64_t f = p->u_first[u], n = p->u_nsess[u];
    const int lane = p->heap_lane[0];           /* shortest lane, lowest index on ties */
    if (lane_push(&p->lanes[lane], f, n) != 0) return -1;
    p->qlen[lane] += n;
    p->heap_len[0] += n;
    heap_down(p, 0);
    (*pointer)++;
    if (*pointer >= n_users) return p->heap_len[0] > p->S ? 2 : 1;   /* 2: the walk ended exactly as the queues filled (the pass is over, nothing more needed) */
  }
}

/* The widest session among the next S of every lane (the batch's Lmax), without popping. */
int64_t pk_peek_lmax(void* h) {
  packer_t* p = (packer_t*)h;
  int64_t m = 0;
  for (int b = 0; b < p->B; b++)
    for (int s = 0; s < p->S; s++) {
      const int64_t len = p->s_len[p->lanes[b].v[p->lanes[b].head + s]];
      if (len > m) m = len;
    }
  return m;
}

/* Pops S sessions per lane; sess [B][S] receives their numbers (the slot-list protocol builds its float64 arrays from them);
 * x, y [B][S][Lmax] int32, mask [B][S] float, slot_len [S] (the reference's per-slot widths) unless x is NULL. */
void pk_pop(void* h, int64_t* sess, int32_t* x, int32_t* y, float* mask, int32_t* slot_len, int64_t Lmax) {
  packer_t* p = (packer_t*)h;
  const int B = p->B, S = p->S;
  for (int b = 0; b < B; b++) {
    lane_t* q = &p->lanes[b];
    for (int s = 0; s < S; s++) sess[(int64_t)b * S + s] = q->v[q->head + s];
    q->head += S;
    p->qlen[b] -= S;
  }
  /* every key drops by S: the heap order is unchanged */
  for (int i = 0; i < B; i++) p->heap_len[i] -= S;
  if (!x) return;
  for (int s = 0; s < S; s++) {
    int64_t w = 0;
    for (int b = 0; b < B; b++) {
      const int64_t len = p->s_len[sess[(int64_t)b * S + s]];
      if (len > w) w = len;
    }
    slot_len[s] = (int32_t)w;
  }
  memset(x, 0, (size_t)B * S * Lmax * sizeof(int32_t));
  memset(y, 0, (size_t)B * S * Lmax * sizeof(int32_t));
  for (int b = 0; b < B; b++)
    for (int s = 0; s < S; s++) {
      const int64_t id = sess[(int64_t)b * S + s];
      const int64_t len = p->s_len[id], beg = p->s_begin[id];
      int32_t* yr = y + ((int64_t)b * S + s) * Lmax;
      int32_t* xr = x + ((int64_t)b * S + s) * Lmax;
      for (int64_t t = 0; t < len; t++) yr[t] = (int32_t)p->ids[beg + t];
      /* x = y shifted by one step, zero past the slot's own width (a session shorter than its slot keeps its last id as the input of the
         step behind its end, as the NumPy form does: x[:, :, 1:] = y[:, :, :-1], then x *= t < slot_len) */
      const int64_t w = slot_len[s];
      for (int64_t t = 1; t < w && t <= len; t++) xr[t] = yr[t - 1];
      mask[(int64_t)b * S + s] = (float)p->s_mask[id];
    }
}

int64_t pk_queue_len(void* h, int lane) { return ((packer_t*)h)->qlen[lane]; }
