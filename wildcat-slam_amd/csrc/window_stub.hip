// placeholder until window.hip lands
#include "ctx.h"
void wc_window_free(wc_ctx *) {}
