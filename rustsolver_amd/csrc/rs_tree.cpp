// rs_tree.cpp -- host-side public game tree: the [action_node] axis of the info-set table.
// Rebuilds what build_game_tree (tree_builder.rs:9-143) produces from Options, using the betting
// state machine of state.rs:86-212, so that table shapes and ActionNode.index numbering match the
// Rust solver's; or adopts a tree the Rust side already built (rs_tree_from_nodes).
// No GPU code here: the branchy tree work stays on the host by design.
#include <cstring>
#include <new>

#include "rs_internal.hpp"

namespace {

constexpr double kAllinThreshold = 0.67;  // constants.rs:2 (Options.all_in_threshold is ignored: state.rs:140,145)
constexpr int kMaxRaises = 2;             // constants.rs:5

// Rust `f64 as u32`: saturating, NaN -> 0
uint32_t to_u32(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 4294967295.0) return 4294967295u;
    return static_cast<uint32_t>(x);
}

struct Seat {
    uint32_t stack = 0, wager = 0;
    bool folded = false;
};

// GameState (state.rs:43-51)
class Betting {
  public:
    Seat seat[2];
    uint32_t pot = 0;
    int raises = 0;
    int to_act = 0;
    int street = 0;  // 0 flop, 1 turn, 2 river
    bool settled = false;

    bool someone_folded() const { return seat[0].folded || seat[1].folded; }          // state.rs:87-94
    bool someone_allin() const { return seat[0].stack == 0 || seat[1].stack == 0; }   // state.rs:100-107
    bool hand_over() const { return street == 2 || someone_allin() || someone_folded(); }  // state.rs:95-99

    struct Move {
        int kind;
        double amt;
    };

    // state.rs:125-157; the returned order defines the child order of every action node
    std::vector<Move> moves(const rs_options &opt, int round_idx) const {
        const Seat &me = seat[to_act], &opp = seat[1 - to_act];
        std::vector<Move> out;
        const bool facing_bet = opp.wager > me.wager;
        if (opp.wager == 0) out.push_back({RS_ACT_CHECK, 0.0});
        if (facing_bet) {
            out.push_back({RS_ACT_CALL, 0.0});
            out.push_back({RS_ACT_FOLD, 0.0});
        }
        if (opp.wager == 0) {
            for (int i = 0; i < opt.n_bet_sizes[round_idx]; ++i) {
                const double frac = opt.bet_sizes[round_idx][i];
                out.push_back({RS_ACT_BET, frac});
                if (frac * static_cast<double>(pot) > kAllinThreshold * static_cast<double>(me.stack)) break;
            }
        }
        if (raises < kMaxRaises && !someone_allin() && facing_bet) {
            for (int i = 0; i < opt.n_raise_sizes[round_idx]; ++i) {
                const double mult = opt.raise_sizes[round_idx][i];
                out.push_back({RS_ACT_RAISE, mult});
                if (mult * static_cast<double>(opp.wager) > kAllinThreshold * static_cast<double>(me.stack)) break;
            }
        }
        return out;
    }

    // state.rs:158-212
    Betting after(const Move &m) const {
        Betting n = *this;
        Seat &me = n.seat[n.to_act];
        const Seat &opp_before = seat[1 - to_act];
        switch (m.kind) {
        case RS_ACT_BET:
        case RS_ACT_RAISE: {
            const bool is_bet = m.kind == RS_ACT_BET;
            uint32_t chips = to_u32((is_bet ? static_cast<double>(n.pot) : static_cast<double>(opp_before.wager)) * m.amt);
            if (chips > to_u32(static_cast<double>(me.stack) * kAllinThreshold)) chips = me.stack;
            me.stack -= chips;
            if (is_bet) me.wager = chips;
            else {
                me.wager += chips;
                n.raises += 1;
            }
            n.pot += chips;
            n.to_act = 1 - n.to_act;
            break;
        }
        case RS_ACT_CALL: {
            const uint32_t owed = opp_before.wager - me.wager;
            if (me.stack >= owed) {
                n.pot += owed;
                me.stack -= owed;
            } else {
                n.pot += me.stack;
                me.stack = 0;
            }
            n.settled = true;
            break;
        }
        case RS_ACT_CHECK:
            if (n.to_act == RS_MAX_PLAYERS - 1) n.settled = true;
            n.to_act = 1 - n.to_act;
            break;
        case RS_ACT_FOLD:
            me.folded = true;
            n.pot -= opp_before.wager - me.wager;
            n.settled = true;
            break;
        }
        return n;
    }

    // state.rs:108-124
    Betting next_street() const {
        Betting n = *this;
        n.settled = false;
        n.to_act = 0;
        n.seat[0].wager = n.seat[1].wager = 0;
        n.street = street + 1;
        return n;
    }
};

class Builder {
  public:
    explicit Builder(const rs_options &o) : opt_(o) {}
    rs_tree *tree = new rs_tree();

    void run(const Betting &start) {
        const int root = add(-1, RS_NODE_PRIVATE_CHANCE);  // tree_builder.rs:60-66
        link(root, grow_action(root, 0, start));
    }

  private:
    const rs_options &opt_;

    int add(int parent, int kind) {  // Tree::create_node, tree.rs:48-53
        rs_tree_node nd;
        std::memset(&nd, 0, sizeof(nd));
        nd.kind = kind;
        nd.parent = parent;
        nd.index = -1;
        tree->nodes.push_back(nd);
        return static_cast<int>(tree->nodes.size()) - 1;
    }
    void link(int parent, int child) {
        rs_tree_node &p = tree->nodes[parent];
        p.children[p.n_children++] = child;
    }

    // tree_builder.rs:67-90: index is taken BEFORE the children are built (pre-order)
    int grow_action(int parent, int round_idx, const Betting &st) {
        const int id = add(parent, RS_NODE_ACTION);
        tree->nodes[id].player = static_cast<uint8_t>(st.to_act);
        tree->nodes[id].round_idx = static_cast<uint8_t>(round_idx);
        tree->nodes[id].index = tree->n_action_nodes++;
        for (const Betting::Move &m : st.moves(opt_, round_idx)) {
            const Betting nx = st.after(m);  // tree_builder.rs:94
            int child;
            if (!nx.settled) child = grow_action(id, round_idx, nx);
            else if (nx.hand_over()) child = leaf(id, nx);
            else child = deal(id, round_idx, nx.next_street());
            link(id, child);
            rs_tree_node &me = tree->nodes[id];
            me.action_kind[me.n_children - 1] = m.kind;  // an.actions.push(action), tree_builder.rs:109-114
            me.action_amt[me.n_children - 1] = m.amt;
        }
        return id;
    }

    // tree_builder.rs:116-133
    int leaf(int parent, const Betting &st) {
        const int id = add(parent, RS_NODE_TERMINAL);
        rs_tree_node &nd = tree->nodes[id];
        nd.value = st.pot;
        nd.last_to_act = static_cast<uint8_t>(st.to_act);
        nd.round = st.street;
        nd.ttype = RS_TERM_SHOWDOWN;
        if (st.someone_allin() && st.street != 2) nd.ttype = RS_TERM_ALLIN;
        if (st.someone_folded()) nd.ttype = RS_TERM_UNCONTESTED;
        return id;
    }

    // tree_builder.rs:134-143
    int deal(int parent, int round_idx, const Betting &st) {
        const int id = add(parent, RS_NODE_PUBLIC_CHANCE);
        tree->nodes[id].round = st.street;
        link(id, grow_action(id, round_idx + 1, st));
        return id;
    }
};

}  // namespace

extern "C" {

int rs_options_default(rs_options *out) {  // options.rs:52-81
    if (!out) return rs::fail(RS_ERR_INVALID, "rs_options_default: out is NULL");
    std::memset(out, 0, sizeof(*out));
    out->stack_sizes[0] = out->stack_sizes[1] = 500;
    out->starting_pot = 35;
    out->n_board_cards = 5;  // "4d5dAs3cKs"
    out->n_rounds = 1;
    out->n_bet_sizes[0] = 2;
    out->bet_sizes[0][0] = 0.5;
    out->bet_sizes[0][1] = 1.0;
    out->n_raise_sizes[0] = 1;
    out->raise_sizes[0][0] = 3.0;
    return RS_OK;
}

int rs_tree_build(const rs_options *o, rs_tree **out) {
    if (!o || !out) return rs::fail(RS_ERR_INVALID, "rs_tree_build: NULL argument");
    if (o->n_board_cards < 3 || o->n_board_cards > 5)
        return rs::fail(RS_ERR_INVALID, "invalid board mask");  // state.rs:64 panic
    const int first_street = o->n_board_cards - 3;
    if (o->n_rounds < 1 || o->n_rounds > RS_MAX_ROUNDS)
        return rs::fail(RS_ERR_INVALID, "rs_tree_build: n_rounds must be 1..RS_MAX_ROUNDS");
    // Check-Check always reaches the next street, where the Rust builder indexes bet_sizes[round_idx]
    // (state.rs:138): missing entries are an index panic there, an error here.
    if (first_street + o->n_rounds < 3)
        return rs::fail(RS_ERR_OOB, "rs_tree_build: bet_sizes missing for a later street (Rust: index panic)");
    for (int r = 0; r < o->n_rounds; ++r)
        if (o->n_bet_sizes[r] < 0 || o->n_bet_sizes[r] > RS_MAX_SIZES || o->n_raise_sizes[r] < 0 ||
            o->n_raise_sizes[r] > RS_MAX_SIZES)
            return rs::fail(RS_ERR_INVALID, "rs_tree_build: too many bet / raise sizes");
    Betting start;  // state.rs:53-72
    start.seat[0].stack = o->stack_sizes[0];
    start.seat[1].stack = o->stack_sizes[1];
    start.pot = o->starting_pot;
    start.street = first_street;
    Builder b(*o);
    try {
        b.run(start);
    } catch (const std::bad_alloc &) {
        delete b.tree;
        return rs::fail(RS_ERR_OOM, "rs_tree_build: out of memory");
    }
    for (const rs_tree_node &nd : b.tree->nodes)
        if (nd.n_children > RS_MAX_ACTIONS) {
            delete b.tree;
            return rs::fail(RS_ERR_UNSUPPORTED, "rs_tree_build: more than RS_MAX_ACTIONS actions at a node");
        }
    *out = b.tree;
    return RS_OK;
}

int rs_tree_from_nodes(const rs_tree_node *nodes, int n_nodes, rs_tree **out) {
    if (!nodes || !out || n_nodes <= 0) return rs::fail(RS_ERR_INVALID, "rs_tree_from_nodes: bad argument");
    rs_tree *t = new (std::nothrow) rs_tree();
    if (!t) return rs::fail(RS_ERR_OOM, "rs_tree_from_nodes: out of memory");
    t->nodes.assign(nodes, nodes + n_nodes);
    int n_act = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const rs_tree_node &nd = t->nodes[i];
        bool ok = nd.kind >= RS_NODE_PRIVATE_CHANCE && nd.kind <= RS_NODE_TERMINAL && nd.n_children >= 0 &&
                  nd.n_children <= RS_MAX_ACTIONS && nd.parent < i;
        for (int k = 0; ok && k < nd.n_children; ++k) ok = nd.children[k] > i && nd.children[k] < n_nodes;
        if (nd.kind == RS_NODE_ACTION) {
            ok = ok && nd.player < 2 && nd.round_idx < RS_MAX_ROUNDS && nd.index >= 0;   // zero actions is possible (state.rs:125-157)
            ++n_act;
        }
        if (nd.kind == RS_NODE_TERMINAL) ok = ok && nd.n_children == 0 && nd.last_to_act < 2;
        if ((nd.kind == RS_NODE_PRIVATE_CHANCE || nd.kind == RS_NODE_PUBLIC_CHANCE)) ok = ok && nd.n_children == 1;
        if (!ok) {
            delete t;
            return rs::fail(RS_ERR_INVALID, "rs_tree_from_nodes: malformed node " + std::to_string(i));
        }
    }
    // ActionNode.index must be a permutation of 0..n_act
    std::vector<char> seen(n_act, 0);
    for (const rs_tree_node &nd : t->nodes)
        if (nd.kind == RS_NODE_ACTION) {
            if (nd.index >= n_act || seen[nd.index]) {
                delete t;
                return rs::fail(RS_ERR_INVALID, "rs_tree_from_nodes: ActionNode.index is not a permutation");
            }
            seen[nd.index] = 1;
        }
    t->n_action_nodes = n_act;
    *out = t;
    return RS_OK;
}

void rs_tree_destroy(rs_tree *tree) { delete tree; }

int rs_tree_n_nodes(const rs_tree *tree) { return tree ? static_cast<int>(tree->nodes.size()) : RS_ERR_INVALID; }
int rs_tree_n_action_nodes(const rs_tree *tree) { return tree ? tree->n_action_nodes : RS_ERR_INVALID; }

int rs_tree_get_node(const rs_tree *tree, int node_id, rs_tree_node *out) {
    if (!tree || !out) return rs::fail(RS_ERR_INVALID, "rs_tree_get_node: NULL argument");
    if (node_id < 0 || node_id >= static_cast<int>(tree->nodes.size()))
        return rs::fail(RS_ERR_OOB, "rs_tree_get_node: node id out of bounds");
    *out = tree->nodes[node_id];
    return RS_OK;
}

}  // extern "C"
