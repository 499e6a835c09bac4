// kern_part9.hip -- instantiates part 9 of the engine's kernel classes (see poa_kern_tables.hip.h); compiled in parallel with
// the other parts by smoothxg_amd/build.py and linked into libsxgpoa.so.
#define SXG_KERN_PART 9
#include "poa_kern_tables.hip.h"
