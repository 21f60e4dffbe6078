#pragma once
// stand-in: mrpt/core/initializer.h
#define MRPT_INITIALIZER(f)                     \
    static void f();                            \
    namespace { struct f##_runner { f##_runner() { f(); } } f##_instance; } \
    static void f()
