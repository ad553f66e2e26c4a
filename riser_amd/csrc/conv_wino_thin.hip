// The thin-launch forms of the F(2,3) kernel (conv_wino.hip, conv_wino_kernel.hpp): staging loads one item ahead, for the
// eight-wave small tiles and for the four-wave workgroups of launches with fewer tiles than CUs.  A translation unit of its own
// so that these 33 instantiations compile next to the 70 of conv_wino.hip.
#include "conv_wino_kernel.hpp"

namespace rs {
namespace {

struct Thin {
    int wm, wn, mt, nt;
    KernelFn fn[3];        // chunk = 16, 20, 24
};
#define RS_THIN(WM, WN, MT, NT)                                                                                            \
    {WM, WN, MT, NT, {conv_wino_kernel<WM, WN, MT, NT, 16, false, true>, conv_wino_kernel<WM, WN, MT, NT, 20, false, true>, \
                      conv_wino_kernel<WM, WN, MT, NT, 24, false, true>}}
const Thin kThin[] = {
    RS_THIN(8, 1, 1, 2), RS_THIN(4, 2, 1, 2), RS_THIN(2, 4, 1, 2), RS_THIN(2, 4, 1, 1),
    RS_THIN(4, 1, 1, 1), RS_THIN(2, 2, 1, 1), RS_THIN(4, 1, 1, 2), RS_THIN(2, 2, 1, 2), RS_THIN(1, 4, 1, 2),
    RS_THIN(4, 1, 1, 3), RS_THIN(2, 2, 1, 3),
};
#undef RS_THIN

}  // namespace

KernelFn conv_wino_thin_fn(int wm, int wn, int mt, int nt, int ki) {
    for (const Thin& t : kThin)
        if (t.wm == wm && t.wn == wn && t.mt == mt && t.nt == nt && ki >= 0 && ki < 3) return t.fn[ki];
    return nullptr;
}

}  // namespace rs
