// scene_yaml.hpp -- a reader for the YAML subset the scene files use, and the number
// formatting of the writer.  yaml-cpp, which the reference links (includes/scene.hpp:21), is an
// un-vendored submodule, so the schema of includes/scene.hpp:38-97 is parsed by hand:
// block maps with plain keys, block sequences (including the compact "- - x" form), flow
// sequences "[a, b, c]", plain scalars and '#' comments.
#pragma once

#include <string>
#include <vector>

namespace rt {
namespace yaml {

struct Node {
    enum Kind { Null, Scalar, Sequence, Map } kind = Null;
    std::string scalar;                                // Scalar
    std::vector<Node> items;                           // Sequence
    std::vector<std::pair<std::string, Node>> fields;  // Map, in file order

    bool is_sequence() const { return kind == Sequence; }
    size_t size() const { return kind == Sequence ? items.size() : kind == Map ? fields.size() : 0; }
    const Node* find(const std::string& key) const;
};

// Parses `text`; on failure returns false and sets `error` ("line N: ...").
bool parse(const std::string& text, Node* out, std::string* error);

// Scalar conversions with yaml-cpp's as<float>() / as<int>() strictness: the whole scalar must
// be consumed.
bool to_float(const Node& n, float* out);
bool to_int(const Node& n, int* out);

// Shortest decimal that reads back to the same float, printed the way an ostream prints with
// that precision ("%.{p}g"): 10 -> "10", 0.5672846f -> "0.5672846".
std::string format_float(float v);

}  // namespace yaml
}  // namespace rt
