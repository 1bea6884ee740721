/* qhip_bench.h — benchmark support, NOT part of the drop-in boundary (that is include/qhip.h / libqhip.so): the synthetic
 * TPC-H-shaped table generators of SURVEY §8(d) that bench.py, tools/bench_host.cpp and the full-size tests fill their inputs with,
 * and the plain streaming-read yardstick. Built into qurious_amd/libqhip_bench.so (csrc/synth.cpp, csrc/bench_stream.hip); nothing
 * in libqhip.so depends on it. Return codes: 0 = ok, non-zero = bad arguments / a HIP error. */
#ifndef QHIP_BENCH_H
#define QHIP_BENCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- measurement aid (SURVEY §8d "achievable-copy ceiling") */
/* Reads `bytes` of a scratch HBM buffer of device `device` with a plain 16-byte-per-lane streaming kernel (non-temporal loads, the fastest
 * variant found on MI355X; `iters` timed launches after one warm-up) and returns the achieved read bandwidth in GB/s: the practical ceiling the filter+aggregate kernel's
 * roofline fraction can be compared with, next to the 8 TB/s data-sheet peak. */
int qhip_bench_stream_read(int32_t device, int64_t bytes, int32_t iters, double* gb_per_s);

/* ---------------------------------------------------------------- synthetic TPC-H-shaped inputs */
/* Counter-based generators (splitmix64, seed 0x515552494F555301) writing Arrow-layout host
 * buffers the caller allocated. Row i of every column depends only on (seed, column, i). */
int qhip_synth_lineitem(int64_t first_row, int64_t n_rows,
                        int32_t* l_shipdate, int32_t* l_returnflag_offsets, uint8_t* l_returnflag_data,
                        int32_t* l_linestatus_offsets, uint8_t* l_linestatus_data,
                        void* l_quantity /* i128[] */, void* l_extendedprice, void* l_discount, void* l_tax);
/* Q3 tables: customers first_key.. (c_custkey 1-based; seg_data holds 10*n bytes), orders with ordinal first_k.. (TPC-H's
 * sparse o_orderkey, o_custkey uniform over 1..n_customers), and the 1..7 lineitems of those orders in order. */
int qhip_synth_customer(int64_t first_key, int64_t n, int64_t* c_custkey, int32_t* seg_offsets, uint8_t* seg_data);
int qhip_synth_orders(int64_t first_k, int64_t n, int64_t n_customers, int64_t* o_orderkey, int64_t* o_custkey,
                      int32_t* o_orderdate, int64_t* o_shippriority);
int64_t qhip_synth_q3_lineitem_count(int64_t first_k, int64_t n_orders);
int qhip_synth_q3_lineitem(int64_t first_k, int64_t n_orders, int64_t* l_orderkey, int32_t* l_shipdate,
                           void* l_extendedprice, void* l_discount);

#ifdef __cplusplus
}
#endif
#endif /* QHIP_BENCH_H */
