#pragma once
// stand-in: mrpt/typemeta/TEnumType.h (string <-> enum registration macros)
#define MRPT_ENUM_TYPE_BEGIN_NAMESPACE(NS, T)
#define MRPT_ENUM_TYPE_BEGIN(T)
#define MRPT_FILL_ENUM(v)
#define MRPT_FILL_ENUM_MEMBER(T, v)
#define MRPT_ENUM_TYPE_END()
