/* oracle/orc_slic.c -- superpixel refinement of the masks.
 * TEST INFRASTRUCTURE ONLY (see orc.h).  Filled in after the main path. */
#include "orc.h"
#include "orc_internal.h"

int orc_superpixel_refine(orc_t* o, const uint8_t* rgb, const uint16_t* depth, uint8_t* masks, int nm, int frame)
{
    (void)o; (void)rgb; (void)depth; (void)masks; (void)nm; (void)frame;
    return 0;
}
