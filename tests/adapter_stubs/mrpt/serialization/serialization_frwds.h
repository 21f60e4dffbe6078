#pragma once
namespace mrpt::serialization { class CArchive; }
