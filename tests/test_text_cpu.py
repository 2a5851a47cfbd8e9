"""Host logic of the paragraph query grammar (no GPU): the cases of nidx_paragraph/src/query_parser/tokenizer.rs's own
tests (test_empty_query, test_simple_query, quotes / exclusions / unclosed quotes) and query_parser.rs's stop-word test
shape."""
import pytest

from nucliadb_amd.text import facet_ancestors, is_valid_facet, parse_query


def lit(*words):
    return [("literal", w) for w in words]


def test_empty_queries():
    for q in ("", "    ", "  - - -   - - -  ", '"  "', "!@#~&/()=?"):
        assert parse_query(q) == [], q


def test_literals_quotes_and_exclusions():
    assert parse_query("This is a simple query") == lit("this", "is", "a", "simple", "query")
    assert parse_query('some "exact match" here') == [("literal", "some"), ("quoted", "exact match"), ("literal", "here")]
    assert parse_query("keep -drop this") == [("literal", "keep"), ("excluded", "drop"), ("literal", "this")]
    # punctuation inside a literal splits it; the dash inside a word is not an exclusion
    assert parse_query("do-stuff a.b") == lit("do", "stuff", "a", "b")
    # an unclosed quote is dropped, the words stay
    assert parse_query('shoupd + enaugh"') == lit("shoupd", "enaugh")
    assert parse_query('"unclosed words') == lit("unclosed", "words")
    assert parse_query('"Quoted, With! Punctuation"') == [("quoted", "quoted with punctuation")]
    assert parse_query("some ' document") == lit("some", "document")


def test_stop_words_keep_the_last_token():
    stop = {"is", "a", "for", "the", "and"}
    assert parse_query("nuclia is a database for unstructured data", stop) == lit("nuclia", "database", "unstructured", "data")
    assert parse_query("nuclia is a database for the", stop) == lit("nuclia", "database", "the")
    assert parse_query("is a for and", stop) == lit("and")
    assert parse_query('nuclia "is" a database', stop) == [("literal", "nuclia"), ("quoted", "is"), ("literal", "database")]


def test_facet_paths():
    assert facet_ancestors("/l/labelset/label") == ["/l", "/l/labelset", "/l/labelset/label"]
    assert is_valid_facet("/l") and is_valid_facet("/e/PERSON") and not is_valid_facet("") and not is_valid_facet("l/x")


def test_adapt_text_quote_fixing():
    """nidx_text/tests/test_search.rs:291-310 (test_quote_fixing) on TextReaderService::adapt_text (reader.rs:357-365)."""
    from nucliadb_amd.text import TextSearcher

    for body in ('"enough test"', 'enough test"', '"enough test'):
        assert TextSearcher.adapt_text(body) == '"enough test"'
    assert TextSearcher.adapt_text("") == "" and TextSearcher.adapt_text("enough test") == "enough test"
    assert TextSearcher.adapt_text('a "b c" d "e') == '"a b c d e"'


def test_tantivy_grammar_subset_parses_like_the_query_parser():
    from nucliadb_amd.text import QuerySyntaxError, flatten_conjunction, parse_text_query

    def shape(q):
        m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query(q))
        assert not not_subs and not must_subs
        return [l.text for l in m], [l.text for l in n], [[l.text for l in grp] for grp in g]

    assert shape("enough to test") == (["enough", "to", "test"], [], [])
    assert shape('"enough to test" more') == (["enough to test", "more"], [], [])
    assert shape("+a -b c") == (["a", "c"], ["b"], [])
    assert shape("(a OR b) AND NOT c") == ([], ["c"], [["a", "b"]])
    assert shape("a AND b OR c AND d".replace(" OR ", " AND ")) == (["a", "b", "c", "d"], [], [])
    assert shape("x (a OR b) (c OR d)") == (["x"], [], [["a", "b"], ["c", "d"]])
    assert shape("NOT (a OR b) z") == (["z"], ["a", "b"], [])
    m = flatten_conjunction(parse_text_query("text:foo^2.5 *"))[0]
    assert (m[0].text, m[0].boost, m[1].all) == ("foo", 2.5, True)
    # nested shapes (round 3): an AND inside an OR is a group member that is a conjunction node, a negated conjunction and a
    # boosted conjunction are sub-trees; a boost on an OR group is carried by its members
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("a OR b c"))
    assert (m, n, not_subs, must_subs) == ([], [], [], []) and g[0][0].text == "a" and g[0][1][0].op == "and" and [l.text for l in g[0][1][0].children] == ["b", "c"]
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("z NOT (a AND b)"))
    assert [l.text for l in m] == ["z"] and [l.text for l in not_subs[0].children] == ["a", "b"]
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("(a b)^2.5 c"))
    assert [l.text for l in m] == ["c"] and must_subs[0][1] == 2.5
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("(a OR b)^2 c"))
    assert [(l.text, l.boost) for l in g[0]] == [("a", 2.0), ("b", 2.0)]
    for bad in ['"enough test', "enough test\"", "a AND", "a OR", "(a b", "a b)", "enough - test", "title:x", "a^"]:
        with pytest.raises(QuerySyntaxError):
            parse_text_query(bad)
    # phrase slop (round 4): `"a b"~2` is a PhraseQuery with set_slop(2); a boost may follow
    leaf = parse_text_query('"a b"~2^3')
    assert (leaf.text, leaf.slop, leaf.boost) == ("a b", 2, 3.0) and parse_text_query('"a b"').slop == 0
    with pytest.raises(QuerySyntaxError):
        parse_text_query('"a b"~')
    # deeper trees (round 4): every level of the tree flattens the same way, what nests below it is a (node, boost) member or a sub-tree
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("x (a OR (b (c OR (d e))))"))
    inner = g[0][1][0]   # b (c OR (d e))
    assert [l.text for l in m] == ["x"] and g[0][0].text == "a" and inner.op == "and"
    m2, n2, g2, ns2, ms2 = flatten_conjunction(inner)
    assert [l.text for l in m2] == ["b"] and g2[0][0].text == "c" and [l.text for l in g2[0][1][0].children] == ["d", "e"]
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("z NOT (a OR (b c))"))
    assert [l.text for l in m] == ["z"] and not_subs[0].op == "or"
    m, n, g, not_subs, must_subs = flatten_conjunction(parse_text_query("a OR (b OR (c d))"))   # an OR inside an OR: one group
    assert g[0][0].text == "a" and g[0][1].text == "b" and g[0][2][0].op == "and"
    # ranges over the text field's terms: inclusive / exclusive / open ends, a field prefix, a boost; bounds are single tokens
    from nucliadb_amd.text import Vocabulary, terms_in_range

    leaf = flatten_conjunction(parse_text_query("x text:[Apple TO melon}^2"))[0][1]
    assert leaf.term_range == ("apple", True, "melon", False) and leaf.boost == 2.0
    boosted = flatten_conjunction(parse_text_query("([a TO b])^3 OR (c [d TO e])^2"))
    assert boosted[2][0][0].term_range == ("a", True, "b", True) and boosted[2][0][0].boost == 3.0
    assert parse_text_query("{a TO *]").term_range == ("a", False, None, True) and parse_text_query('["b" TO c]').term_range == ("b", True, "c", True)
    for bad in ["[a b TO c]", "[a TO ]", "[a c]", "[a TO b", '["a b" TO c]']:
        with pytest.raises(QuerySyntaxError):
            parse_text_query(bad)
    vocab = Vocabulary()
    ids = {t: vocab.id(t) for t in ["apple", "banana", "cherry", "melon", "zebra", "\x00label:/l/x", "émile"]}
    pick = lambda *r: [t for t in ids if ids[t] in terms_in_range(vocab, *r)]
    assert pick("apple", True, "melon", False) == ["apple", "banana", "cherry"] and pick("apple", False, "melon", True) == ["banana", "cherry", "melon"]
    assert pick(None, True, "b", True) == ["apple"] and pick("n", True, None, True) == ["zebra", "émile"] and pick("x", True, "a", True) == []


def test_deletion_terms_respect_seq_and_key_kind():
    from nucliadb_amd.text import Vocabulary, deletion_terms

    v = Vocabulary()
    rid = "%032x" % 7
    a, b = v.id("\x00uuid:" + rid), v.id("\x00fid:" + rid + "/a/body")
    dele = [(rid, 5), (rid + "/a/body", 3), ("%032x" % 9, 9)]
    assert deletion_terms(v, 2, dele) == sorted([a, b])
    assert deletion_terms(v, 3, dele) == [a]          # `Seq(segment) < del_seq` is strict
    assert deletion_terms(v, 5, dele) == []


def test_tricky_resource_of_the_reference_at_the_host_level():
    """nidx_paragraph/tests/reader.rs:71-91 (create_tricky_resource) and :452-500 (test_query_parsing_weird_stuff), :418-450
    (test_query_parsing): what those cases pin on the host side — the document tokenizer ("default": split on non-alphanumerics,
    lower-case), the keyword grammar and the stop-word rule (every stop word but the last token goes; the lists are the
    reference's own data files, read from /root/reference when it is there)."""
    import glob
    import json
    import os

    from nucliadb_amd.bm25 import tokenize

    paragraphs = ["That's a too *tricky* resource", "It's very important to do-stuff", "It's not that important to do-stuff",
                  "W'h'a't a -w-e-i-r-d p\"ara\"gra\"ph"]
    tokens = [tokenize(p) for p in paragraphs]
    matching = lambda word: sum(word in t for t in tokens)   # noqa: E731
    # `total` of the one-word queries of the reference's test
    assert (matching("important"), matching("paragraph"), matching("ara"), matching("ph"), matching("to")) == (2, 0, 1, 1, 2)
    assert tokens[3][-4:] == ["p", "ara", "gra", "ph"] and tokens[1][-2:] == ["do", "stuff"]

    def contains_phrase(doc, words):
        return any(doc[i:i + len(words)] == words for i in range(len(doc) - len(words) + 1))

    quoted = parse_query('"It\'s very important to do-stuff"')
    assert quoted == [("quoted", "it s very important to do stuff")]
    assert sum(contains_phrase(t, quoted[0][1].split()) for t in tokens) == 1            # `total` == 1
    assert parse_query("some ' document") == lit("some", "document")                      # the stray quote goes (PR 3216)

    base = "/root/reference/nidx/nidx_paragraph/stop_words"
    if not os.path.isdir(base):
        pytest.skip("the reference's stop-word data files are not on this machine")
    stop = set()
    for code in ("fr", "it", "es", "en", "ca", "de", "nl", "pt"):   # LOADED_LANGUAGES, query_parser/stop_words.rs:84-93
        with open(os.path.join(base, code + ".json")) as f:
            stop |= set(json.load(f))
    assert len(glob.glob(os.path.join(base, "*.json"))) >= 8
    # "removes all stop words except the last and matches `to` exactly": ematches == ["to"]
    assert parse_query("it's not that to", stop) == lit("to")
    assert parse_query("some ' document", stop) == lit("document")
    assert parse_query("important", stop) == lit("important")


class _StubIndex:
    """What TextSearcher / ParagraphSearcher need of an index to map a request to clauses (no device)."""

    def __init__(self, words):
        from nucliadb_amd.text import ALL_DOCS, NOT_REPEATED, Vocabulary

        self.vocab = Vocabulary()
        for w in list(words) + [ALL_DOCS, NOT_REPEATED]:
            self.vocab.id(w)
        self.empty_term = len(self.vocab.ids)

    def term(self, w):
        t = self.vocab.lookup(w)
        return self.empty_term if t is None else t


def _shape(c):
    """A clause tree as nested tuples: ("sub", occur, [children]) / ("phrase", occur, slop) / ("set", occur, n) / (occur,)"""
    if c.subquery is not None:
        return ("sub", c.occur, [_shape(l) for l in c.subquery])
    if c.term_set is not None and c.phrase:
        return ("phrase", c.occur, c.slop)
    if c.term_set is not None:
        return ("set", c.occur, len(c.term_set))
    return (c.occur,)


def test_every_parsed_body_and_formula_maps_to_a_clause_tree():
    """Round 4: nothing tantivy's QueryParser parses (nidx_text/src/reader.rs:357-376) and no nesting of a filtering formula
    (nidx_paragraph/src/search_query.rs:88-143) is refused any more — the host mapping on a stub index: the tree of nested
    BooleanQuerys each request becomes."""
    from nucliadb_amd import _lib
    from nucliadb_amd.text import (DocumentSearchRequest, FormulaLiteral as L, FormulaNot, FormulaOp, ParagraphSearcher, ParagraphSearchRequest,
                                   TextSearcher)

    M, S, N, G = _lib.OCCUR_MUST, _lib.OCCUR_SHOULD, _lib.OCCUR_MUST_NOT, _lib.OCCUR_SHOULD_GROUP
    ts = TextSearcher(_StubIndex(list("abcdefgh")))
    tree = lambda body: [_shape(c) for c in ts._clauses(DocumentSearchRequest(body=body))]
    assert tree("a OR (b (c OR (d e)))") == [(G,), ("sub", G, [(M,), (G,), ("sub", G, [(M,), (M,)])])]
    assert tree("a ((b OR c) (d OR e))") == [(M,), (G,), (G,), (G + 1,), (G + 1,)]                  # a conjunction flattens into its level
    assert tree("h OR ((b OR c) (d OR e))") == [(G,), ("sub", G, [(G,), (G,), (G + 1,), (G + 1,)])]   # a nested query without a Must leaf
    assert tree("a NOT (b OR (c d))") == [(M,), ("sub", N, [(G,), ("sub", G, [(M,), (M,)])])]
    assert tree('a OR (b "c d"~2 [e TO g])') == [(G,), ("sub", G, [(M,), ("phrase", M, 2), ("set", M, 3)])]
    assert tree("a OR (NOT b)") == [(G,)] and tree("(a (b c)^2)^3 OR d")[0][0] == "sub"
    nine = " ".join(f"({x} OR h)" for x in "abcdefgh") + " (a OR b)"
    t9 = tree(nine)
    assert [x for x in t9 if x[0] == "sub"] == [("sub", M, [(S,), (S,)])] and len(t9) == 17       # the ninth group nests
    with pytest.raises(ValueError):
        tree("a OR (" + " ".join("abcdefgh"[i % 8] for i in range(33)) + ")")
    ps = ParagraphSearcher(_StubIndex(["\x00label:/a", "\x00label:/b", "\x00label:/c", "\x00label:/d"]))
    form = lambda f, **kw: [_shape(c) for c in ps._filter_query(ParagraphSearchRequest(body="", filtering_formula=f, **kw), None, 1.0)]
    a, b, c, d = L("/a"), L("/b"), L("/c"), L("/d")
    assert form(FormulaOp("or", [FormulaOp("and", [FormulaOp("or", [a, b]), FormulaOp("or", [c, d])]), a])) == \
        [("sub", G + 1, [(G,), (G,), (G + 1,), (G + 1,)]), (G + 1,)]
    assert form(FormulaNot(FormulaOp("or", [FormulaOp("and", [a, b]), c]))) == [(M,), ("sub", N, [("sub", G, [(M,), (M,)]), (G,)])]
    assert form(FormulaNot(FormulaNot(a))) == [(M,), ("sub", N, [(M,), (N,)])]
    deep = FormulaOp("and", [a, FormulaOp("or", [b, FormulaOp("and", [c, FormulaOp("or", [d, FormulaNot(FormulaOp("and", [a, b]))])])])])
    assert form(deep) == [(M,), (G + 1,), ("sub", G + 1, [(M,), (G,), ("sub", G, [(M,), ("sub", N, [(M,), (M,)])])])]
