// ifx_slic.hip -- superpixel refinement of the masks.  Filled in after the main path.
#include "ifx_ctx.h"
#include <vector>
int ifx_superpixel_refine(ifx* h, const uint8_t* rgb, const uint16_t* depth, std::vector<uint8_t>& masks, int nm, int frame)
{
    (void)rgb; (void)depth; (void)masks; (void)nm; (void)frame;
    h->err = "superpixel refinement is not implemented yet";
    return IFX_E_INVALID;
}
