// sphx_knn_group.h - arguments of the lane-per-query grouped search (sphx_knn_group.hip)
#pragma once
#include "sphx_internal.h"

#ifndef KG_TCAP
#define KG_TCAP 1408                     // candidates per group tile after the cull (11 bits of a key name the slot)
#endif
#ifndef KG_RSPREAD
#define KG_RSPREAD 1.5                   // widest radius a group's tile is sized for, in units of the group's smallest
#endif
#define KG_TPRE 3072                     // candidates of the rows of cells before the per-candidate cull (row ids: u16)
#define KG_MAXROWS 512                   // NON-EMPTY rows of cells a group's tile may draw from
#define KG_ROWS_ALL 8192                 // rows of cells (empty ones included) a group's box + radius may span
static_assert(KG_TCAP <= 2048 && KG_TCAP % 64 == 0, "tile slots are named by 11 key bits");

struct KnnGroupArgs {
    int n, k, npad;
    int n_active;              // queries with id >= n_active (ghosts) are skipped
    const double *x, *y, *z;   // cell-sorted positions
    const int* id;
    const int* qorder;         // processing order (nullable: identity)
    const int* cell_start;
    GridParams g;
    const double* rsearch;     // previous h: sorted order, or by id (hint_by_id)
    int hint_by_id;
    double rscale, rbound;
    int* nbr;                  // [k][npad]
    double* h_sorted;          // one of the two
    double* h_by_id;
    int4* tie_list;            // near ties between two consecutive ranks, left to the list-mode launch's tie blocks (nullable: such queries fail over)
    int* tie_count;
    int tie_cap;
    int* fail_list;            // processing slots this kernel could not certify
    int* fail_count;
    u64* counters;
    int exp_noamb;             // timing experiment (SPHX_KG_EXP_NOAMB): 0 in the product
    u64* prof;                 // diagnostics (nullptr in the product): per-section cycle sums
};
int sphx_knn_group(sphx_ctx* ctx, const KnnGroupArgs& a);
