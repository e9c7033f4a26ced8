#include "models.h"
Model* dmx_make_unet(const dmx_unet_config*) { dmx_set_error("unet not built yet"); return nullptr; }
