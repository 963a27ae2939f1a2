/* host_tables.h -- host-libm constant tables shipped to the device (see host_tables.cpp). */
#ifndef SRLA_HOST_TABLES_H
#define SRLA_HOST_TABLES_H

#include <stdint.h>

#include "device_layout.h"

namespace srla {

uint32_t complex_table_len(uint32_t m);
/* entries (double2) of the twiddle block of one FFT size:
 * [complex fwd stages][complex inv stages][real fwd, nfft/4][real inv, nfft/4] */
uint32_t twiddle_count(uint32_t nfft);
void build_twiddles(uint32_t nfft, double *out);
void fill_geom(uint32_t n, SrlaGeom *g);
void build_rice_thresholds(double *thr);

}  // namespace srla
#endif
