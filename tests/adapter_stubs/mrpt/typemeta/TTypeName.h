#pragma once
#define DECLARE_TTYPENAME_CLASSNAME(x)
