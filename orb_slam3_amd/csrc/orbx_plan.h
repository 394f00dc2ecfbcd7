// Host/device shared description of one extraction plan: everything the
// kernels need to know about an (image size, extractor parameters, batch
// capacity) combination.  Built once on the host (orbx_api.hip), copied to
// device memory, and read through wave-uniform loads.
#ifndef ORBX_PLAN_H_
#define ORBX_PLAN_H_

#include <stddef.h>
#include <stdint.h>

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19          /* EDGE_THRESHOLD, core/lib/ORB/extractor.cc:73 */
#define ORBX_DET_ORG 16       /* minBorder = EDGE_THRESHOLD - 3, extractor.cc:754 */
#define ORBX_CELL_MAX 72      /* cell side never exceeds 70 (= 2*35) pixels */
#define ORBX_MAX_ROOTS 16

namespace orbx {

// Resize tap table entry, indexed by PADDED destination coordinate (the
// REFLECT_101 border mapping is folded in): two source indices and the two
// 11-bit fixed-point weights of cv::resize INTER_LINEAR.
struct alignas(8) Tap {   // 8-byte aligned: wave-uniform entries are fetched with scalar loads
  uint16_t s0, s1;
  int16_t w0, w1;
};

// Per destination dword column (4 padded pixels) of a resized level: where its
// source bytes start and how the 4 pixels pick from the 8 bytes that follow.
struct ColRec {
  uint16_t sxa;      // source x of pixel 0's left tap
  uint16_t deltas;   // 3 bits per pixel: left tap - sxa (0..6); bit 15: fast path valid
  uint16_t w1[4];    // right-tap weight per pixel (left = 2048 - w1)
  uint16_t nvalid;   // pixels of this dword inside the padded width (1..4); the rest store 0
  uint16_t pad;
};

struct LevelDesc {
  int w, h;              // level image size
  int pstride, prows;    // padded plane: row stride (bytes, multiple of 64), rows = h + 38
  size_t pyr_off;        // byte offset of frame 0's padded plane in the pyramid buffer
  size_t pyr_plane;      // bytes per frame
  int bstride;           // blurred image row stride (multiple of 64)
  size_t blur_off, blur_plane;
  // detection grid (extractor.cc:762-768); only cells with a non-empty domain
  int ncx, ncy, wCell, hCell;
  int cell_first;        // first cell of this level in the per-frame numbering
  int cand_cap_cell;     // candidate slots per cell = ceil(wCell/2)*ceil(hCell/2)
  uint32_t cand_off;     // uint32 offset of this level's cell segments inside a frame
  uint32_t flat_off;     // uint32 offset of this level's flat key area inside a frame
  uint32_t flat_cap;
  // blur tiling
  int tiles_x, tiles_y, tile_first;
  // quadtree (extractor.cc:549-568)
  int quota;             // mnFeaturesPerLevel[level]
  int n_roots;
  float hX;
  int sel_cap, sel_off;  // selected-key slots for this level inside a frame
  float scale;           // mvScaleFactor[level]
  int patch_size;        // (int)(31 * scale)
  const Tap* xtab;       // [w + 38]
  const Tap* ytab;       // [h + 38]
  // k_pyr_resize: one column record per destination dword; the extents below describe the
  // source window of a 256 x 16 destination tile (kept for a tile-staging variant of the kernel;
  // the current kernel fetches its taps straight from the source plane and does not read them)
  const ColRec* colrec;  // [pstride / 4]
  const uint16_t* xblk;  // per 256-column block: lowest / highest source x touched
  const uint16_t* yblk;  // per 16-row block: lowest / highest source y touched
  const uint16_t* bcols; // dword columns that need the per-pixel tap path (borders)
  int n_bcols;
  int rs_ls;             // row stride (bytes) such a staged source window would need
  int rs_rows;           // rows such a window would need
};

// What k_fast_cells needs to know about one detection cell of a frame (all levels, in cell order).
struct FastCell {
  uint32_t xy;       // X0 | Y0 << 16: level coordinates of the cell's first pixel
  uint32_t geo;      // width | height << 8 | level << 16 | load shift << 20 | (row stride / 64) << 24
  uint32_t cand;     // the cell's first slot in a frame's candidate buffer
  uint32_t tile;     // byte offset, inside the level's padded plane, of the aligned start of the tile's first row
  uint32_t pyr64;    // the level's planes in the pyramid buffer, / 64
  uint32_t plane64;  // bytes per plane, / 64
  uint32_t magic;    // 65536 / (quick-test tasks per row), rounded up: row = lane * magic >> 16
  uint32_t pad;
};

struct Plan {
  int nlevels, W, H;
  int ini_th, min_th;
  int max_batch;
  int cells_per_frame;
  int tiles_per_frame;
  uint32_t cand_frame;   // uint32 per frame in the per-cell candidate buffer
  uint32_t flat_frame;   // uint32 per frame in each flat key buffer
  int sel_frame;         // selected-key slots per frame (all levels)
  int kp_cap;            // keypoint slots per frame in the outputs (= sel_frame)
  // k_fast_cells LDS geometry (max over levels): row stride, tile rows, m-map rows,
  // survivor queue entries, bit-mask words
  int fast_ts, fast_tile_rows, fast_mm_rows, fast_qcap, fast_words;
  // 1: detect at iniThFAST first and at minThFAST only in cells that found nothing; 0: one pass
  // at the lower threshold that serves both (same results; kept switchable for timing)
  int fast_two_pass;
  const FastCell* fast_cells;   // [cells_per_frame]
  // 1: the descriptor kernel blurs each keypoint's own window in LDS and no blurred plane is
  // produced; 0: a full-plane blur launch feeds a descriptor kernel that reads blurred planes
  // (kept for comparison and for the blurred-plane stage dump; same results)
  int blur_in_describe;
  int blur_waves;        // waves per workgroup of the blur kernel, 4 or 1 (timing knob)
  int pyr_waves;         // waves per workgroup of the pyramid kernels (timing knob; results do not change)
  int pyr_xcd;           // 1: pyramid workgroups renumbered so that neighbours share an XCD's L2 (timing knob)
  int oct_nodes_cap;     // quadtree node slots (max over levels)
  int oct_cells_max;     // max cells in one level
  int oct_keys_lds;      // keys per LDS key buffer of k_octree (0 = always use global memory)
  int oct_team_keys;     // keys per LDS key buffer of k_octree_team (0 = kernel not used)
  int oct_team_frames;   // batches of at most this many frames build their trees with k_octree_team
  uint32_t oct_ws_bytes; // 0: node arrays in LDS; else bytes of global workspace per (frame, level)
  int umax[16];          // circular patch row half-widths (extractor.cc:448-462)
  unsigned long long umax_packed;   // umax[v] in bits 4 v .. 4 v + 3
  // per patch row v = j - 15 (j = 0 .. 30; row 31 empty): 0xff in byte b of the row's 32 bytes
  // where pixel u = b - 15 lies inside the circular patch, |u| <= umax[|v|] (extractor.cc:75-99)
  uint32_t ic_mask[32][8];
  LevelDesc lv[ORBX_MAX_LEVELS];
};

// One unit of work for the orientation + descriptor kernel: where the keypoint's two windows
// start (so that the kernel's first loads depend on nothing but this record) and where its
// results go.
struct WorkItem {
  // byte offsets of the keypoint's windows.  Blur inside the descriptor kernel: pyr_win = the
  // 43-row window of the pyramid plane (row cy - 21, the dword that holds column cx - 21),
  // blur_win unused.  Blurred planes: pyr_win = the 31-row patch window (row cy - 15, the dword
  // that holds column cx - 15), blur_win = the 37-row sample window in the blurred buffer (row
  // cy - 18, the dword that holds column cx - 18).
  uint64_t pyr_win;
  uint64_t blur_win;
  uint32_t xy;        // x | y << 16, level coordinates
  int dst;            // output slot inside the frame
  uint32_t strides;   // row stride of the level's padded plane | row stride of its blurred plane << 16
  int level;
};

}  // namespace orbx

#endif
