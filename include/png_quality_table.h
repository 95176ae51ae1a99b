/* png_quality_table.h -- the one constant table the lossy PNG row shares between the device build's host code and the oracle */
#ifndef PNG_QUALITY_TABLE_H
#define PNG_QUALITY_TABLE_H
#include <stdint.h>
/* -q on a PNG: libimagequant turns the quality into a largest acceptable mean square error and uses the fewest colours that reach it
   (quality_to_mse: 2.5 / (210 + q)^1.2 * (100.1 - q) / 100 plus a fudge below q ~ 15 [UPSTREAM-RECALL]).  Restated over this quantiser:
   the cut stops as soon as the error of the palette it would give -- bin means against box means, squared 8-bit levels summed over the
   four channels, plain sRGB values rather than imagequant's gamma-weighted ones -- is within that bound; q 100 cuts to 256 colours, q 0
   stops at two.  The table is the bound times 255^2 * 1024, per pixel. */
static const uint64_t kQualityBound[101] = {
    0, 1265794, 729701, 548126, 455282, 397967, 358441, 329099, 306139, 287446,
    271754, 258253, 246407, 235841, 226290, 217558, 209501, 205916, 202368, 198856,
    195380, 191939, 188532, 185160, 181822, 178517, 175244, 172004, 168796, 165620,
    162474, 159359, 156275, 153220, 150194, 147198, 144230, 141291, 138379, 135495,
    132638, 129807, 127004, 124226, 121474, 118747, 116046, 113369, 110717, 108089,
    105485, 102904, 100347, 97812, 95301, 92812, 90344, 87899, 85475, 83073,
    80692, 78331, 75991, 73671, 71372, 69092, 66832, 64591, 62369, 60167,
    57983, 55817, 53670, 51540, 49429, 47335, 45258, 43199, 41157, 39131,
    37123, 35130, 33154, 31194, 29250, 27322, 25409, 23512, 21629, 19762,
    17910, 16072, 14249, 12441, 10646, 8866, 7100, 5347, 3608, 1883,
    0};
#endif
