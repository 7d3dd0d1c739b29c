// dsact_tu_fat.hip -- kernel family "fat" of libdsact.so: explicit instantiations only (see dsact_tu.h)
#include "dsact_tu.h"
#define DSACT_K_chain_fwd DSACT_SKIP
#define DSACT_K_chain_pipe DSACT_SKIP
#define DSACT_K_chain_bwd DSACT_SKIP
#define DSACT_K_chain_merged DSACT_SKIP
#define DSACT_K_fat DSACT_INSTANTIATE
#define DSACT_K_conv DSACT_SKIP
#define DSACT_K_tiles DSACT_SKIP
#include "dsact_instances.inc"
