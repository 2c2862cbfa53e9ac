"""Exhaustive interleavings of the hand-over of a cut chain (smfft_amd/csrc/smfft_kernels.hpp: ChainState, chain_own, chain_park,
chain_resume_or_take) on a small model: two workgroups -- the OWNER of the chain's head and its RESUMER -- each a sequence of atomic
steps on one shared word and one tile of d_output, scheduled in every possible order, the resumer's time-out free to fire at any of
its polls (or never).  Whatever the order: the chain's `nreuses` applications are applied exactly once each, in order, to the data
the chain started from; nobody reads the parked tile before it is complete; nothing is stored over a result; and the model never
waits for a workgroup that has not started (a resumer that sees `Owned` waits only because the owner is running).  CPU only: this
pins the PROTOCOL; that the kernels implement it is what the GPU tests with a delayed owner check.  (The kernel's resumer reads the
word and compare-and-swaps the value it read in two steps; a swap that fails because the owner's max came in between changes
nothing and is, for the model, a poll that saw OWNED.)"""
import itertools

NOTHING, OWNED, PARKED, TAKEN = 0, 1, 2, 3      # the word, relative to the launch's base (anything below OWNED: an earlier launch's)


class World:
    def __init__(self, nreuses, cut, stale):
        self.word = stale                       # whatever an earlier launch left: always below this launch's OWNED
        self.nreuses, self.cut = nreuses, cut   # the owner runs applications [0, cut), the resumer [cut, nreuses)
        self.d_input = ("x", 0)                 # (data, applications applied)
        self.d_output = None                    # the chain's slot of d_output
        self.output_final = False
        self.log = []

    def copy(self):
        w = World(self.nreuses, self.cut, self.word)
        w.__dict__.update({k: (list(v) if isinstance(v, list) else v) for k, v in self.__dict__.items()})
        return w


def owner_steps():
    """generator protocol: each yield is one atomic step; it receives the world and returns None (continue) or 'done'"""
    def own(w, st):
        old = w.word
        w.word = max(w.word, OWNED)             # ONE atomic max
        st["go"] = old != TAKEN
        return None if st["go"] else "done"     # taken: the head is not computed, nothing is stored

    def compute(w, st):
        data, apps = w.d_input
        assert apps == 0
        st["tile"] = (data, w.cut)              # applications [0, cut) in LDS
        return None

    def store(w, st):
        assert not w.output_final, "the owner stored over a finished chain"
        w.d_output = st["tile"]                 # write-through stores, drained before the word is set
        return None

    def park(w, st):
        assert w.word == OWNED, w.word          # nobody may have changed a word its owner holds
        w.word = PARKED
        return "done"
    return [own, compute, store, park]


def resumer_poll(w, st, timed_out):
    """one poll of the resumer; returns None (poll again) or 'resume' / 'take'"""
    v = w.word
    if v == PARKED:
        return "resume"
    if v != OWNED and timed_out:
        # compare-and-swap of the stale value it polled: atomic with the read in this model step
        w.word = TAKEN
        return "take"
    return None


def resumer_finish(w, how):
    if how == "resume":
        assert w.d_output is not None, "the resumer read a tile nobody parked"
        data, apps = w.d_output
        assert apps == w.cut, "the resumer read an incomplete or foreign tile"
    else:
        data, apps = w.d_input                  # the whole chain from d_input
        assert apps == 0
    w.d_output = (data, w.nreuses)
    w.output_final = True


def explore(nreuses, cut, stale, max_polls=4):
    """all schedules: a schedule is a sequence of choices 'o' (owner's next step) / 'r' (a resumer poll without time-out) /
    'R' (a resumer poll at which its time-out has expired)"""
    outcomes = set()
    osteps = owner_steps()

    def rec(w, oi, ost, rdone, polls):
        owner_done = oi == "done"
        if owner_done and rdone:
            outcomes.add((w.d_output, w.word))
            return
        if not owner_done:
            w2, st2 = w.copy(), dict(ost)
            res = osteps[oi](w2, st2)
            rec(w2, "done" if res == "done" else oi + 1, st2, rdone, polls)
        if not rdone:
            choices = (False, True) if polls < max_polls else (True,)
            for timed_out in choices:
                w2 = w.copy()
                how = resumer_poll(w2, {}, timed_out)
                if how is None:
                    # polling again is only progress if something else can still move: the owner (a resumer that sees OWNED waits for a RUNNING owner)
                    if owner_done:
                        assert w2.word in (PARKED,), "the resumer would wait for ever"     # unreachable: PARKED returns 'resume'
                    if not owner_done and polls < max_polls + 8:
                        rec(w2, oi, ost, False, polls + 1)
                    continue
                resumer_finish(w2, how)
                rec(w2, oi, ost, True, polls)

    rec(World(nreuses, cut, stale), 0, {}, False, 0)
    return outcomes


def test_every_interleaving_applies_every_application_once():
    for nreuses, cut, stale in itertools.product((3, 7), (1, 2), (NOTHING, -5)):
        outcomes = explore(nreuses, cut, stale)
        assert outcomes, (nreuses, cut)
        for d_output, word in outcomes:
            assert d_output == ("x", nreuses), (d_output, word, nreuses, cut, stale)
            assert word in (PARKED, TAKEN), word
        # both endings are reachable: the handed-over chain and the taken one
        assert {w for _, w in outcomes} == {PARKED, TAKEN}


def test_a_resumer_never_takes_a_chain_whose_owner_has_started():
    """once the owner's atomic max has happened the word is OWNED or PARKED: a time-out cannot turn it into TAKEN"""
    w = World(5, 2, NOTHING)
    st = {}
    assert owner_steps()[0](w, st) is None and w.word == OWNED
    assert resumer_poll(w, {}, True) is None and w.word == OWNED
    for step in owner_steps()[1:]:
        step(w, st)
    assert w.word == PARKED and resumer_poll(w, {}, True) == "resume"


def test_an_owner_that_starts_late_leaves_a_taken_chain_alone():
    w = World(5, 2, -9)
    assert resumer_poll(w, {}, True) == "take"
    resumer_finish(w, "take")
    st = {}
    assert owner_steps()[0](w, st) == "done" and not st["go"]
    assert w.d_output == ("x", 5) and w.word == TAKEN
