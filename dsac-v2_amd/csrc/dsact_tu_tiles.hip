// dsact_tu_tiles.hip -- kernel family "tiles" of libdsact.so: explicit instantiations only (see dsact_tu.h)
#include "dsact_tu.h"
#define DSACT_K_chain_fwd DSACT_SKIP
#define DSACT_K_chain_pipe DSACT_SKIP
#define DSACT_K_chain_bwd DSACT_SKIP
#define DSACT_K_chain_merged DSACT_SKIP
#define DSACT_K_fat DSACT_SKIP
#define DSACT_K_conv DSACT_SKIP
#define DSACT_K_tiles DSACT_INSTANTIATE
#include "dsact_instances.inc"
