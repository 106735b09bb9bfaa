// TEST DOUBLE — see plugin.h in this directory.  Csound's <modload.h> includes <plugin.h> and defines
// the module entry points (csoundModuleCreate / csoundModuleInit / csoundModuleDestroy) that call
// csnd::on_load(); the tests call csnd::on_load() themselves.
#pragma once
#include "plugin.h"
