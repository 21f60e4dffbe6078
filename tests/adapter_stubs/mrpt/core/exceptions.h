#pragma once
// stand-in: mrpt/core/exceptions.h (THROW_EXCEPTION*, ASSERT_*)
#include <stdexcept>
#include <string>
#define THROW_EXCEPTION(msg) throw std::runtime_error(std::string(msg))
#define THROW_EXCEPTION_FMT(fmt, ...) throw std::runtime_error(std::string(fmt))
#define ASSERT_(c) do { if (!(c)) throw std::runtime_error("assert: " #c); } while (0)
#define ASSERTMSG_(c, msg) do { if (!(c)) throw std::runtime_error(std::string(msg)); } while (0)
#define ASSERT_GT_(a, b) ASSERT_((a) > (b))
#define ASSERT_GE_(a, b) ASSERT_((a) >= (b))
#define ASSERT_LE_(a, b) ASSERT_((a) <= (b))
#define ASSERT_LT_(a, b) ASSERT_((a) < (b))
#define MRPT_THROW_UNKNOWN_SERIALIZATION_VERSION(v) throw std::runtime_error("unknown serialization version")
#define MRPT_START
#define MRPT_END
